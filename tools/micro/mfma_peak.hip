// Microbenchmark: what does the fp32 MFMA pipe deliver on this chip in the forward kernel's occupancy
// regime (512 workgroups x 4 waves, 2 waves per SIMD)?  Pure register-operand MFMA chains, no LDS, no HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS> void run(const char* name, int blocks, int mfma_per_wave) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int iters = mfma_per_wave / (16 * CHAINS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<CHAINS><<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) k<CHAINS><<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double flop = (double)blocks * 4 * iters * 16 * CHAINS * 4096.0;
    printf("%s blocks=%d mfma/wave=%d: %.1f us  %.1f TFLOP/s\n", name, blocks, iters * 16 * CHAINS, ms * 1e3, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>("1 chain ", 512, 2560);
    run<4>("4 chains", 512, 2560);
    run<1>("1 chain ", 256, 5120);     // one wave per SIMD
    run<4>("4 chains", 256, 5120);
    run<1>("1 chain x10", 512, 25600);
    return 0;
}
