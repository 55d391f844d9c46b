// Microbenchmark + exactness check: the residuals of the bf16x3 operand split by v_dot2c_f32_bf16 (a - p.lo = dot2(p, {-1, 0}) + a)
// instead of shift/and + v_sub_f32: 7 instead of 9 VALU per element pair.  Checks bit equality of the three terms against the
// shift/sub form over random bit patterns (all exponents) and times both forms (VALU-only loop, 8 waves per CU).
// build: hipcc -O3 --offload-arch=gfx950 split_dot2.hip -o split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { f32x2v v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ void split_shift(float a, float b, unsigned* o) {
    const unsigned p1 = pk(a, b);
    a -= __builtin_bit_cast(float, p1 << 16); b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
    const unsigned p2 = pk(a, b);
    a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
    o[0] = p1; o[1] = p2; o[2] = pk(a, b);
}
__device__ __forceinline__ void split_dot2(float a, float b, unsigned* o) {
    const bf16x2 m0 = {(__bf16)-1.0f, (__bf16)0.0f}, m1 = {(__bf16)0.0f, (__bf16)-1.0f};
    const unsigned p1 = pk(a, b);
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p1), m0, a, false);
    b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p1), m1, b, false);
    const unsigned p2 = pk(a, b);
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p2), m0, a, false);
    b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p2), m1, b, false);
    o[0] = p1; o[1] = p2; o[2] = pk(a, b);
}
__global__ void check(const float* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned s[3], d[3];
    split_shift(in[2 * i], in[2 * i + 1], s);
    split_dot2(in[2 * i], in[2 * i + 1], d);
    for (int j = 0; j < 3; ++j) { out[6 * i + j] = s[j]; out[6 * i + 3 + j] = d[j]; }
}
template <int MODE>
__global__ __launch_bounds__(512, 1) void bench(float* out, int iters, float a0) {
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a0 * (j + 1) + threadIdx.x;
    unsigned sink = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            unsigned o[3];
            if (MODE == 0) split_shift(v[2 * q], v[2 * q + 1], o); else split_dot2(v[2 * q], v[2 * q + 1], o);
            sink ^= o[0] ^ o[1] ^ o[2];
            v[2 * q] += 1.5f; v[2 * q + 1] += 2.5f;
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)sink;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(2 * n);
    srand(7);
    for (int i = 0; i < 2 * n; ++i) {
        unsigned u = ((unsigned)rand() << 17) ^ ((unsigned)rand() << 2) ^ (unsigned)rand();
        if (i % 3 == 0) { float f = (float)rand() / RAND_MAX * 8.0f - 4.0f; memcpy(&u, &f, 4); }   // a third in the usual range
        memcpy(&h[i], &u, 4);
    }
    float* din; unsigned* dout;
    (void)hipMalloc(&din, 2 * n * 4); (void)hipMalloc(&dout, 6 * n * 4);
    (void)hipMemcpy(din, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
    check<<<n / 256, 256>>>(din, dout, n);
    std::vector<unsigned> o(6 * n);
    (void)hipMemcpy(o.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        bool neq = false;
        for (int j = 0; j < 3; ++j) neq |= o[6 * i + j] != o[6 * i + 3 + j];
        if (!neq) continue;
        ++bad;
        // is either input a NaN / inf / near the top of the range / below 2^-100 (residuals subnormal)?
        bool special = false;
        for (int c = 0; c < 2; ++c) { unsigned u; memcpy(&u, &h[2 * i + c], 4); const unsigned e = (u >> 23) & 0xff; if (e >= 0xfe || e < 27) special = true; }
        if (!special) { ++bad_normal; if (shown++ < 8) printf("  MISMATCH a=%a b=%a  shift %08x %08x %08x  dot2 %08x %08x %08x\n", h[2 * i], h[2 * i + 1], o[6 * i], o[6 * i + 1], o[6 * i + 2], o[6 * i + 3], o[6 * i + 4], o[6 * i + 5]); }
    }
    printf("pairs %d  differing %ld  of which with both inputs in [2^-100, 2^127): %ld\n", n, bad, bad_normal);
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 100; ++w) { if (mode == 0) bench<0><<<256, 512>>>(out, 2000, 0.5f); else bench<1><<<256, 512>>>(out, 2000, 0.5f); }
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) { if (mode == 0) bench<0><<<256, 512>>>(out, 2000, 0.5f); else bench<1><<<256, 512>>>(out, 2000, 0.5f); }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch (2000 iterations x 8 pairs, 2 waves per SIMD)\n", mode == 0 ? "shift/sub residuals" : "v_dot2c residuals ", ms / 20 * 1e3);
    }
    return 0;
}
