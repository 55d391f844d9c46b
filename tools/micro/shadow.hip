// Microbenchmark: how many VALU instructions of the SAME wave issue in the shadow of one MFMA on gfx950, as a function of the
// MFMA shape, of how the accumulators alternate (1 = one dependent chain, 2 / 4 = round robin over independent accumulators)
// and of the VALU instruction type.  Group = 1 MFMA + K VALU (independent registers), order pinned; one wave per SIMD
// (WAVES = 4) or two (8).  Prints cycles per group: T(K) = max(T_mfma, issue + 4K) if the VALU hide, T_mfma + 4K if not.
// build: hipcc -O3 --offload-arch=gfx950 shadow.hip -o shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VT> __device__ __forceinline__ void valu(float& x, float& y, float& z) {
    if (VT == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (VT == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
    if (VT == 2) { typedef float f2 __attribute__((ext_vector_type(2))); f2 p = {x, y}; asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p)); x = p[0]; y = p[1]; }
    if (VT == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(z) : "v"(x));
    if (VT == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(z) : "v"(x));
}
// independent forms (destination registers rotate over 8, sources are loop invariant): what ONE instruction of each type costs
// in the shadow, without the write-after-write chains of the forms above
typedef float f2 __attribute__((ext_vector_type(2)));
template <int VT> __device__ __forceinline__ void valu_ind(float& d, f2& d2, float y, float w, f2 y2) {
    if (VT == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(y), "v"(w));
    if (VT == 11) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(y));
    if (VT == 12) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(y));
    if (VT == 13) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2) : "v"(y2), "v"(y2));
    if (VT == 14) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(y), "v"(w));
    if (VT == 15) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(y));
    if (VT == 16) asm volatile("v_max_f32 %0, 0, %1" : "=v"(d) : "v"(y));
    if (VT == 17) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(d) : "v"(y), "v"(w));
    if (VT == 18) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(y), "v"(w));      // mask from a register: no literal
}
template <int SHAPE, int ACCS, int K, int VT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, int iters, float a0, unsigned long long* clk) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + (threadIdx.x & 63) * 1e-3f + j); b[j] = (__bf16)(a0 - j * 0.01f); }
    f32x4 c4[4]; f32x16 c16[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 4; ++r) c4[i][r] = 0.f; for (int r = 0; r < 16; ++r) c16[i][r] = 0.f; }
    float x[8], y = a0, z = 0.f;
    for (int j = 0; j < 8; ++j) x[j] = a0 * j + threadIdx.x;
    f2 x2[8], y2 = {a0, a0 + 1.f}; float w = a0 * 3.f;
    for (int j = 0; j < 8; ++j) { x2[j][0] = 0.f; x2[j][1] = 0.f; }
    asm volatile("" : "+v"(y), "+v"(w), "+v"(y2));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            if (SHAPE == 16) c4[g % ACCS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[g % ACCS], 0, 0, 0);
            else             c16[g % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[g % ACCS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < K; ++v) { if constexpr (VT < 10) valu<VT>(x[v & 7], y, z); else valu_ind<VT>(x[(g * K + v) & 7], x2[(g * K + v) & 7], y, w, y2); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && threadIdx.x == 0) clk[0] = t1 - t0;
    float s = z + y;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 4; ++r) s += c4[i][r]; for (int r = 0; r < 16; ++r) s += c16[i][r]; }
    for (int j = 0; j < 8; ++j) s += x[j] + x2[j][0] + x2[j][1];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}
static float* g_out; static unsigned long long* g_clk;
// whole-launch time (HIP events) per group and SIMD: with two waves per SIMD the older wave runs ahead, so wave 0's own
// cycle count does not say what the pair costs
template <int SHAPE, int ACCS, int K, int VT, int WAVES> double run_total() {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 100; ++w) k<SHAPE, ACCS, K, VT, WAVES><<<256, 64 * WAVES>>>(g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) k<SHAPE, ACCS, K, VT, WAVES><<<256, 64 * WAVES>>>(g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e6 / (iters * 8.0) / (WAVES / 4);      // ns per group (one MFMA + K VALU) of one wave, per SIMD
}
template <int SHAPE, int ACCS, int VT, int WAVES> void row_total(const char* vt) {
    printf("%dx%d  accumulators %d  waves/SIMD %d  VALU %-10s K=0..6:", SHAPE, SHAPE, ACCS, WAVES / 4, vt);
    printf(" %5.2f", run_total<SHAPE, ACCS, 0, VT, WAVES>()); printf(" %5.2f", run_total<SHAPE, ACCS, 1, VT, WAVES>());
    printf(" %5.2f", run_total<SHAPE, ACCS, 2, VT, WAVES>()); printf(" %5.2f", run_total<SHAPE, ACCS, 3, VT, WAVES>());
    printf(" %5.2f", run_total<SHAPE, ACCS, 4, VT, WAVES>()); printf(" %5.2f", run_total<SHAPE, ACCS, 5, VT, WAVES>());
    printf(" %5.2f", run_total<SHAPE, ACCS, 6, VT, WAVES>());
    printf("   ns of launch time per group and SIMD\n");
}
template <int SHAPE, int ACCS, int K, int VT, int WAVES> double run1() {
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) k<SHAPE, ACCS, K, VT, WAVES><<<256, 64 * WAVES>>>(g_out, iters, 0.5f, g_clk);
    (void)hipDeviceSynchronize();
    unsigned long long h;
    (void)hipMemcpy(&h, g_clk, 8, hipMemcpyDeviceToHost);
    return (double)h / (iters * 8.0);
}
template <int SHAPE, int ACCS, int VT, int WAVES> void row(const char* vt) {
    printf("%dx%d  accumulators %d  waves/SIMD %d  VALU %-18s K=0..8:", SHAPE, SHAPE, ACCS, WAVES / 4, vt);
    printf(" %5.1f", run1<SHAPE, ACCS, 0, VT, WAVES>()); printf(" %5.1f", run1<SHAPE, ACCS, 1, VT, WAVES>());
    printf(" %5.1f", run1<SHAPE, ACCS, 2, VT, WAVES>()); printf(" %5.1f", run1<SHAPE, ACCS, 3, VT, WAVES>());
    printf(" %5.1f", run1<SHAPE, ACCS, 4, VT, WAVES>()); printf(" %5.1f", run1<SHAPE, ACCS, 5, VT, WAVES>());
    printf(" %5.1f", run1<SHAPE, ACCS, 6, VT, WAVES>()); printf(" %5.1f", run1<SHAPE, ACCS, 8, VT, WAVES>());
    printf("   cycles per group (wave 0 of workgroup 7)\n");
}
template <int SHAPE, int WAVES> void shape() {
    row<SHAPE, 1, 0, WAVES>("v_add_f32"); row<SHAPE, 2, 0, WAVES>("v_add_f32"); row<SHAPE, 4, 0, WAVES>("v_add_f32");
    row<SHAPE, 1, 1, WAVES>("v_cvt_pk_bf16_f32"); row<SHAPE, 2, 1, WAVES>("v_cvt_pk_bf16_f32");
    row<SHAPE, 1, 2, WAVES>("v_pk_add_f32"); row<SHAPE, 2, 2, WAVES>("v_pk_add_f32");
    row<SHAPE, 1, 3, WAVES>("v_exp_f32"); row<SHAPE, 2, 3, WAVES>("v_exp_f32");
    row<SHAPE, 1, 4, WAVES>("v_and_b32"); row<SHAPE, 2, 4, WAVES>("v_and_b32");
}
int main(int argc, char** argv) {
    (void)hipMalloc(&g_out, 256 * 512 * 4); (void)hipMalloc(&g_clk, 16);
    if (argc > 1 && argv[1][0] == 'i') {       // per-type cost in the shadow, whole-launch time, two waves per SIMD
        row_total<16, 2, 0, 8>("v_add_f32"); row_total<16, 2, 10, 8>("cvt_pk_bf16"); row_total<16, 2, 11, 8>("and literal");
        row_total<16, 2, 18, 8>("and reg"); row_total<16, 2, 12, 8>("lshlrev"); row_total<16, 2, 13, 8>("pk_add_f32");
        row_total<16, 2, 14, 8>("sub_f32"); row_total<16, 2, 15, 8>("exp_f32"); row_total<16, 2, 16, 8>("max_f32"); row_total<16, 2, 17, 8>("fma_f32");
        row_total<32, 1, 0, 8>("v_add_f32"); row_total<32, 1, 10, 8>("cvt_pk_bf16"); row_total<32, 1, 11, 8>("and literal");
        row_total<32, 1, 13, 8>("pk_add_f32"); row_total<32, 1, 15, 8>("exp_f32");
        return 0;
    }
    if (argc > 1) {
        row_total<16, 2, 0, 4>("v_add_f32"); row_total<16, 2, 0, 8>("v_add_f32"); row_total<16, 1, 0, 8>("v_add_f32");
        row_total<32, 1, 0, 4>("v_add_f32"); row_total<32, 1, 0, 8>("v_add_f32");
        return 0;
    }
    shape<16, 4>(); shape<32, 4>(); shape<16, 8>(); shape<32, 8>();
    return 0;
}
