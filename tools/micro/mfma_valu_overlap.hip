// Microbenchmark: can the VALU work of one wave run under the bf16 MFMAs of ANOTHER wave of the same SIMD?
// Workgroup = 8 waves (2 per SIMD, wave w and w+4 share one).  Waves 0..3 run a dependent chain of
// v_mfma_f32_32x32x16_bf16, waves 4..7 a VALU loop (the operand-split mix: cvt_pk / shift / and / pk_add).
// Compare: MFMA only, VALU only, both (specialised), and both mixed inside every wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

template <int MODE>   // 0: waves 0-3 MFMA, 4-7 idle; 1: 0-3 idle, 4-7 VALU; 2: specialised; 3: every wave does both (half of each)
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float a0) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(a0 - j); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a0 * (j + 1) + threadIdx.x;
    unsigned sink = 0;
    const bool do_mfma = (MODE == 0 && wave < 4) || (MODE == 2 && wave < 4) || MODE == 3;
    const bool do_valu = (MODE == 1 && wave >= 4) || (MODE == 2 && wave >= 4) || MODE == 3;
    const int n = (MODE == 3) ? iters / 2 : iters;
    for (int i = 0; i < n; ++i) {
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 24; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (do_valu) {   // 8 element pairs x 9 ops = 72 VALU per iteration
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float x = v[2 * q], y = v[2 * q + 1];
                const unsigned p1 = pk(x, y);
                x -= __builtin_bit_cast(float, p1 << 16); y -= __builtin_bit_cast(float, p1 & 0xffff0000u);
                const unsigned p2 = pk(x, y);
                x -= __builtin_bit_cast(float, p2 << 16); y -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                sink ^= p1 ^ p2 ^ pk(x, y);
                v[2 * q] = x + 1.5f; v[2 * q + 1] = y + 2.5f;
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)sink;
}
template <int MODE> float run(const char* name, int iters) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 100; ++w) k<MODE><<<256, 512>>>(out, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) k<MODE><<<256, 512>>>(out, iters, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-44s %8.1f us\n", name, ms * 1e3);
    (void)hipFree(out);
    return ms;
}
int main() {
    const int iters = 400;   // per wave: 9600 MFMA (x33 cycles) or 28800+ VALU
    run<0>("MFMA only (waves 0-3)", iters);
    run<1>("VALU only (waves 4-7)", iters);
    run<2>("specialised: 0-3 MFMA + 4-7 VALU", iters);
    run<3>("mixed: every wave half MFMA + half VALU", iters);
    return 0;
}
