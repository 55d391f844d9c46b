// Microbenchmark: bf16 MFMA (v_mfma_f32_32x32x16_bf16, gfx950) rate in register-operand chains, next to the fp32
// MFMA, for the "error-free bf16 split" study (6 bf16 MFMAs of K=16 replace 8 fp32 MFMAs of K=2 per 32x32x16 tile
// product).  Also with interleaved VALU work (the operand splitting) to see whether the two pipes overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int VALU>
__global__ __launch_bounds__(256, 2) void kb(float* out, int iters, float a0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(a0 - j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a0 + j + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU; ++q) v[q & 7] = v[q & 7] * 1.0001f + v[(q + 1) & 7];
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int VALU> void run(const char* name, int blocks, int mfma_per_wave) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int iters = mfma_per_wave / (16 * CHAINS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) kb<CHAINS, VALU><<<blocks, 256>>>(out, iters, 0.5f);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) kb<CHAINS, VALU><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double flop = (double)blocks * 4 * iters * 16 * CHAINS * 32768.0;
    printf("%s blocks=%d mfma/wave=%d valu/mfma=%.2f: %.1f us  %.1f TFLOP/s bf16  (= %.1f 'fp32-equivalent' TFLOP/s at 6 MFMA per product)\n",
           name, blocks, iters * 16 * CHAINS, (double)VALU / CHAINS, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 6);
    hipFree(out);
}
int main() {
    run<1, 0>("1 chain ", 512, 10240);
    run<2, 0>("2 chains", 512, 10240);
    run<4, 0>("4 chains", 512, 10240);
    run<4, 0>("4 chains 1 wave/SIMD", 256, 20480);
    run<2, 2>("2 chains + VALU", 512, 10240);
    run<2, 4>("2 chains + VALU", 512, 10240);
    run<2, 8>("2 chains + VALU", 512, 10240);
    run<2, 16>("2 chains + VALU", 512, 10240);
    return 0;
}
