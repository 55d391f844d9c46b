// Microbenchmark: how much of a wave's VALU phase hides under its SIMD partner's MFMA phase when the two waves of a SIMD
// run the SAME phase program (barrier; MFMA block; VALU block) half a phase apart ("stagger", MI355X_MICROARCH.md
// "Two waves per SIMD" item 9) -- the structure of lsnf_fwd3.hip's panel loop.
//   workgroup = 8 waves (wave w and w+4 share a SIMD), one workgroup per CU, 256 workgroups.
//   phase = NM x v_mfma_f32_16x16x32_bf16 (two alternating accumulators) + NV VALU of the operand-split mix.
// Modes: 0 MFMA only (all waves)   1 VALU only (all waves)   2 in phase: every wave [barrier, MFMA, VALU]
//        3 staggered: waves 0-3 [barrier, MFMA, VALU], waves 4-7 [barrier, VALU, MFMA]
//        4 staggered + s_setprio 1 on waves 4-7   5 interleaved inside every wave (1 MFMA : NV/NM VALU), no stagger
//        6 = 3 without any barrier (free-running)
// build: hipcc -O3 --offload-arch=gfx950 stagger.hip -o stagger
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef SHAPE32     // NM counts 16-cycle units either way: one 32x32x16 MFMA = two units
typedef f32x16 acc_t;
#define MFMA_PAIR(c0, c1, a, b, b2) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#define NACC 16
#else
typedef f32x4 acc_t;
#define MFMA_PAIR(c0, c1, a, b, b2) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c1, 0, 0, 0);
#define NACC 4
#endif
template <int NM>
__device__ __forceinline__ void mfma_block(acc_t& c0, acc_t& c1, const bf16x8& a, const bf16x8& b, const bf16x8& b2) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NM / 2; ++u) { MFMA_PAIR(c0, c1, a, b, b2) }
    __builtin_amdgcn_sched_barrier(0);
}
// PAIRS element pairs x 9 VALU (cvt_pk, 2 shift/and, 2 sub, cvt_pk, 2 shift/and+sub...) ~ the bf16x3 operand split
// the MFMA block with the wave idling NOP+1 cycles after every MFMA: does a back-to-back MFMA stream hold the SIMD's VALU issue
// port (head-of-line) so that the partner wave's VALU cannot slip in, and do the gaps let it in?
template <int NM, int NOP>
__device__ __forceinline__ void mfma_block_gaps(acc_t& c0, acc_t& c1, const bf16x8& a, const bf16x8& b, const bf16x8& b2) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NM / 2; ++u) {
#ifdef SHAPE32
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" :: "n"(NOP)); asm volatile("s_nop %0" :: "n"(NOP)); __builtin_amdgcn_sched_barrier(0);
#else
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" :: "n"(NOP)); __builtin_amdgcn_sched_barrier(0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" :: "n"(NOP)); __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int PAIRS>
__device__ __forceinline__ void valu_block(float* v, unsigned& sink) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        float x = v[(2 * q) & 15], y = v[(2 * q + 1) & 15];
        const unsigned p1 = pk(x, y);
        x -= __builtin_bit_cast(float, p1 << 16); y -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned p2 = pk(x, y);
        x -= __builtin_bit_cast(float, p2 << 16); y -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        sink ^= p1 ^ p2 ^ pk(x, y);
        v[(2 * q) & 15] = x + 1.5f; v[(2 * q + 1) & 15] = y + 2.5f;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int NM, int PAIRS>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float a0, unsigned long long* clk) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    acc_t c0, c1; for (int r = 0; r < NACC; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 a, b, b2;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + (threadIdx.x & 63) * 1e-3f + j); b[j] = (__bf16)(a0 - j * 0.01f); b2[j] = (__bf16)(a0 + j * 0.02f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a0 * (j + 1) + threadIdx.x;
    unsigned sink = 0;
    if (MODE == 4 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { mfma_block<NM>(c0, c1, a, b, b2); }
        else if (MODE == 1) { valu_block<PAIRS>(v, sink); }
        else if (MODE == 2) { __builtin_amdgcn_s_barrier(); mfma_block<NM>(c0, c1, a, b, b2); valu_block<PAIRS>(v, sink); }
        else if (MODE == 3 || MODE == 4 || MODE == 6) {
            if (MODE != 6) __builtin_amdgcn_s_barrier();
            if (wave < 4) { mfma_block<NM>(c0, c1, a, b, b2); valu_block<PAIRS>(v, sink); }
            else          { valu_block<PAIRS>(v, sink); mfma_block<NM>(c0, c1, a, b, b2); }
        } else if (MODE >= 10 && MODE < 30) {          // staggered, gaps of MODE-10 after every MFMA
            __builtin_amdgcn_s_barrier();
            if (wave < 4) { mfma_block_gaps<NM, MODE - 10>(c0, c1, a, b, b2); valu_block<PAIRS>(v, sink); }
            else          { valu_block<PAIRS>(v, sink); mfma_block_gaps<NM, MODE - 10>(c0, c1, a, b, b2); }
        } else if (MODE >= 30 && MODE < 50) {          // the same + s_setprio 1 around the MFMA phase
            __builtin_amdgcn_s_barrier();
            if (wave < 4) { __builtin_amdgcn_s_setprio(1); mfma_block_gaps<NM, MODE - 30>(c0, c1, a, b, b2); __builtin_amdgcn_s_setprio(0); valu_block<PAIRS>(v, sink); }
            else          { valu_block<PAIRS>(v, sink); __builtin_amdgcn_s_setprio(1); mfma_block_gaps<NM, MODE - 30>(c0, c1, a, b, b2); __builtin_amdgcn_s_setprio(0); }
        } else if (MODE >= 50 && MODE < 70) {          // MFMA only, with the gaps (what the gaps cost by themselves)
            mfma_block_gaps<NM, MODE - 50>(c0, c1, a, b, b2);
        } else if (MODE >= 70 && MODE < 90) {          // in phase [bar, MFMA with gaps + prio, VALU]: the older wave runs ahead by itself
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1); mfma_block_gaps<NM, MODE - 70>(c0, c1, a, b, b2); __builtin_amdgcn_s_setprio(0); valu_block<PAIRS>(v, sink);
        } else if (MODE == 5) {
            __builtin_amdgcn_s_barrier();
            // the same work, interleaved by the scheduler: groups of 1 MFMA + (9*PAIRS/NM) VALU
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NM / 2; ++u) { MFMA_PAIR(c0, c1, a, b, b2) }
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                float x = v[(2 * q) & 15], y = v[(2 * q + 1) & 15];
                const unsigned p1 = pk(x, y);
                x -= __builtin_bit_cast(float, p1 << 16); y -= __builtin_bit_cast(float, p1 & 0xffff0000u);
                const unsigned p2 = pk(x, y);
                x -= __builtin_bit_cast(float, p2 << 16); y -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                sink ^= p1 ^ p2 ^ pk(x, y);
                v[(2 * q) & 15] = x + 1.5f; v[(2 * q + 1) & 15] = y + 2.5f;
            }
#pragma unroll
#ifdef SHAPE32
            for (int u = 0; u < NM / 2; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (14 * PAIRS + NM / 2 - 1) / (NM / 2), 0);
            }
#else
            for (int u = 0; u < NM; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (14 * PAIRS + NM - 1) / NM, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    for (int r = 0; r < NACC; ++r) s += c0[r] + c1[r];
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)sink;
}
template <int MODE, int NM, int PAIRS> float run(const char* name, int iters) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16); unsigned long long h[2];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) k<MODE, NM, PAIRS><<<256, 512>>>(out, iters, 0.5f, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) k<MODE, NM, PAIRS><<<256, 512>>>(out, iters, 0.5f, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("NM=%3d PAIRS=%3d  %-52s %8.1f us   wave cycles %8llu  (%.0f per iter)  clock %.2f GHz\n", NM, PAIRS, name, ms * 1e3, h[0], (double)h[0] / iters, (double)h[0] / (double)h[1] / 10.0);
    (void)hipFree(out);
    return ms;
}
template <int NM, int PAIRS> void suite(int iters) {
    run<0, NM, PAIRS>("MFMA only", iters);
    run<1, NM, PAIRS>("VALU only", iters);
    run<2, NM, PAIRS>("in phase [bar, MFMA, VALU]", iters);
    run<3, NM, PAIRS>("staggered: w0-3 [bar,MFMA,VALU] w4-7 [bar,VALU,MFMA]", iters);
    run<4, NM, PAIRS>("staggered + setprio 1 on w4-7", iters);
    run<5, NM, PAIRS>("interleaved inside every wave", iters);
    run<6, NM, PAIRS>("staggered, no barrier", iters);
}
template <int NM, int PAIRS, int NOP> void gaps(int iters) {
    char nm[96];
    snprintf(nm, sizeof nm, "MFMA only, s_nop %d after each", NOP);            run<50 + NOP, NM, PAIRS>(nm, iters);
    snprintf(nm, sizeof nm, "staggered, s_nop %d gaps", NOP);                  run<10 + NOP, NM, PAIRS>(nm, iters);
    snprintf(nm, sizeof nm, "staggered, s_nop %d gaps, setprio in MFMA", NOP); run<30 + NOP, NM, PAIRS>(nm, iters);
    snprintf(nm, sizeof nm, "in phase, s_nop %d gaps, setprio in MFMA", NOP);  run<70 + NOP, NM, PAIRS>(nm, iters);
}
template <int NM, int PAIRS> void ratio(int iters) {       // in-wave interleave at different VALU : MFMA ratios (14 VALU per pair)
    run<0, NM, PAIRS>("MFMA only", iters);
    run<1, NM, PAIRS>("VALU only", iters);
    run<2, NM, PAIRS>("in phase [bar, MFMA, VALU]", iters);
    run<5, NM, PAIRS>("interleaved inside every wave", iters);
}
int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'r') { ratio<96, 4>(200); ratio<96, 7>(200); ratio<96, 10>(200); ratio<96, 14>(200); return 0; }
    if (argc > 1) {      // the gap experiment
        gaps<96, 16, 0>(200); gaps<96, 16, 1>(200); gaps<96, 16, 3>(200); gaps<96, 16, 5>(200); gaps<96, 16, 7>(200); gaps<96, 16, 9>(200); gaps<96, 16, 11>(200);
        return 0;
    }
    suite<96, 16>(200);     // VALU/MFMA issue-cycle ratio 144*4 / 96*16 = 0.375
    suite<96, 32>(200);     // 0.75
    suite<192, 32>(100);    // 0.375, longer phases
    suite<48, 8>(400);      // 0.375, shorter phases
    return 0;
}
