// Microbenchmark: fp32 MFMA shapes 32x32x2 vs 16x16x4 on RANDOM operands (DVFS: the clock the chip holds under
// load can depend on the shape and on the data, MI355X_MICROARCH.md 'DVFS give-back' items 1, 7).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; }
__device__ unsigned long long stamps[2];
template <bool BIG, bool RANDOM>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = RANDOM ? rnd(s) : 0.f; b[i] = RANDOM ? rnd(s) * 0.1f : 0.f; }
    f32x16 accB[2]; f32x4 accS[8];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) accB[c][r] = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) accS[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (BIG) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                accB[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], accB[0], 0, 0, 0);
                accB[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[15 - u], b[u], accB[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) accS[(u & 1) * 4 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + c) & 15], b[u], accS[(u & 1) * 4 + c], 0, 0, 0);
            }
        }
    }
    float t = 0;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) t += accB[c][r];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) t += accS[c][r];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = t;
    if (blockIdx.x == 7 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
template <bool BIG, bool RANDOM> void run(const char* name, int iters = 4000, int reps = 5) {
    const int blocks = 512;
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    // BIG: 32 MFMA x 4096 flop; SMALL: 64 MFMA x 2048 flop per iteration -> equal flops
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<BIG, RANDOM><<<blocks, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) k<BIG, RANDOM><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double flop = (double)blocks * 4 * iters * 32 * 4096.0;
    unsigned long long hs[2]; (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(stamps), 16);
    printf("%-28s %.2f ms  %.1f TFLOP/s | in-kernel: %llu memtime ticks, %llu realtime ticks (100 MHz) -> %.3f GHz by stamps; ideal pipe cycles %.0f -> %.3f GHz if the pipe never idles\n",
           name, ms, flop / ms / 1e9, hs[0], hs[1], (double)hs[0] / hs[1] * 0.1, (double)iters * 32 * 64 * 2, (double)iters * 32 * 64 * 2 / (ms * 1e6));
    (void)hipFree(out);
}
int main() {
    run<true, true>("32x32x2 random, 2560 MFMA/wave x200 launches", 80, 200);
    run<true, true>("32x32x2 random, 5120 MFMA/wave x200", 160, 200);
    run<true, true>("32x32x2 random, 25600 MFMA/wave x50", 800, 50);
    run<true, false>("32x32x2  zeros");
    run<true, true>("32x32x2  random");
    run<false, false>("16x16x4  zeros");
    run<false, true>("16x16x4  random");
    run<true, true>("32x32x2  random (again)");
    run<false, true>("16x16x4  random (again)");
    return 0;
}
