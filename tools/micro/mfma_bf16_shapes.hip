// Microbenchmark: sustained rate of the two bf16 MFMA shapes on RANDOM operands (the chip lowers its clock under the
// bf16 pipe's load; on zeros / constants it does not): v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16, register
// operands re-read from LDS every 6 MFMAs (the split-bf16 forward's pattern), 8 waves per CU, >= 1 s of launches first.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    __shared__ bf16x8 lds[512 * 6];
    for (int i = threadIdx.x; i < 512 * 6; i += 512) lds[i] = src[RANDOM ? i : 0];
    __syncthreads();
    const bf16x8* wp = lds + (threadIdx.x & ~63) * 6 + (threadIdx.x & 63);
    bf16x8 b[3];
    for (int p = 0; p < 3; ++p) b[p] = src[RANDOM ? 4096 + threadIdx.x * 3 + p : 0];
    f32x16 acc32; f32x4 acc16[4];
    for (int r = 0; r < 16; ++r) acc32[r] = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) acc16[t][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        bf16x8 a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = wp[((i & 1) * 3 + p) * 64];
        if (SHAPE == 32) {
#pragma unroll
            for (int t = 0; t < 6; ++t) acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3], b[t / 2], acc32, 0, 0, 0);
        } else {   // same flops: 12 MFMAs of 16x16x32 on 4 accumulators
#pragma unroll
            for (int t = 0; t < 12; ++t) acc16[t & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % 3], b[(t / 2) % 3], acc16[t & 3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc32[r];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc16[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, bool RANDOM> void run(const char* name, const bf16x8* src, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2500; ++w) k<SHAPE, RANDOM><<<256, 512>>>(src, out, iters);     // ~1 s of load first
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 200; ++w) k<SHAPE, RANDOM><<<256, 512>>>(src, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 200;
    const double flop = 256.0 * 8 * iters * 6 * 32768.0;
    printf("%-46s %7.1f us  %7.1f TFLOP/s\n", name, ms * 1e3, flop / ms / 1e9);
}
int main() {
    const size_t n = 8192 * 8;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    srand(1);
    for (size_t i = 0; i < n; ++i) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; __builtin_memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    bf16x8* src; float* out;
    (void)hipMalloc(&src, n * 2); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    run<32, true>("32x32x16, random operands", src, out);
    run<16, true>("16x16x32, random operands", src, out);
    (void)hipMemset(src, 0, n * 2);
    run<32, false>("32x32x16, zero operands", src, out);
    run<16, false>("16x16x32, zero operands", src, out);
    return 0;
}
