// lsnf_small3.h -- building blocks of the bf16x3 latency kernels (lsnf_small3_fwd.hip, lsnf_small3_bwd.hip): 16-sample
// workgroups on v_mfma_f32_16x16x32_bf16, "L16" lane layout (lsnf_layout.h): lane = (n = lane & 15 -> sample,
// g = lane >> 4); a half-unit (16 features x 16 samples) is 4 registers per lane: feature 16*ft + 4*g + r.
// Stage outputs are final (no K-splitting): the producer splits the 4 values it holds per lane into their three bf16
// terms and stores them in B-operand order; consumers read ready operands (3 ds_read_b128 per 32-feature k-tile).
#pragma once
#include "lsnf_device.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define S3_SAMPLES 16
#define S3_BTILE_FLOATS 768            // one 32-feature k-tile in B-operand order: 3 parts x 64 lanes x 16 B

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
// the 4 values a lane holds of a half-unit -> their three bf16 terms, 2 dwords each (slots ft*4 .. ft*4+3 of the B operand)
__device__ __forceinline__ void split4(const f32x4& x, u32x2* out /*[3]*/) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float a = x[2 * q], b = x[2 * q + 1];
        const unsigned p1 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p1 << 16); b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned p2 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        out[0][q] = p1; out[1][q] = p2; out[2][q] = pk_bf16(a, b);
    }
}
// producer: half-unit ft of B-tile `tile` <- split(x)
__device__ __forceinline__ void store_half(float* tile, int ft, const f32x4& x, int lane) {
    u32x2 p[3];
    split4(x, p);
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x2*>(tile + i * 256 + lane * 4 + ft * 2) = p[i];
}
struct BOp { bf16x8 p[3]; };
__device__ __forceinline__ BOp load_btile(const float* tile, int lane) {
    BOp b;
#pragma unroll
    for (int i = 0; i < 3; ++i) b.p[i] = *reinterpret_cast<const bf16x8*>(tile + i * 256 + lane * 4);
    return b;
}
// weight fragments of one half-unit (nt, ft) of a stage with KT k-tiles
template <int KT> struct UFrags { bf16x8 w[KT][3]; };
template <int KT>
__device__ __forceinline__ UFrags<KT> fetch_unit(const float* stage, int nt, int ft, int lane) {
    UFrags<KT> f;
    const bf16x8* g = reinterpret_cast<const bf16x8*>(stage + (size_t)nt * KT * LSNF_FRAG3_FLOATS) + ft * 3 * 64 + lane;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[kt][p] = g[(kt * 6 + p) * 64];
    return f;
}
#define S3_TERMS(M) M(2, 0) M(0, 2) M(1, 1) M(1, 0) M(0, 1) M(0, 0)
// acc(16 features x 16 samples) += W_unit^T in,  in = KT B-tiles in LDS
template <int KT>
__device__ __forceinline__ f32x4 unit_mma(f32x4 acc, const UFrags<KT>& f, const float* in_tiles, int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const BOp b = load_btile(in_tiles + kt * S3_BTILE_FLOATS, lane);
#define S3_MMA(WI, XI) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[kt][WI], b.p[XI], acc, 0, 0, 0);
        S3_TERMS(S3_MMA)
#undef S3_MMA
    }
    return acc;
}
// bias of half-unit (nt-th bias block at cst, ft): the [h][r] order of the 32x32 layout (lsnf_prep.hip bias_feature)
__device__ __forceinline__ f32x4 unit_bias(const float* cst, int ft, int g) {
    return *reinterpret_cast<const f32x4*>(cst + (g & 1) * 16 + 4 * (2 * ft + (g >> 1)));
}
__device__ __forceinline__ f32x4 relu4(f32x4 a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
    return a;
}
// latent rows <-> half-units: tile t of the split-pad row, half-unit ft
template <int HT>
__device__ __forceinline__ f32x4 load_row_half(int t, int ft, const float* __restrict__ zr, int half, int g, int vw) {
    const int hh = t / HT, tt = t % HT, f0 = 32 * tt + 16 * ft + 4 * g, col0 = hh * half + f0;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vw == 4) {
        if (f0 < half) v = *reinterpret_cast<const f32x4*>(zr + col0);
    } else if (vw == 2) {
        if (f0 < half) { const f32x2 a = *reinterpret_cast<const f32x2*>(zr + col0); v[0] = a[0]; v[1] = a[1]; }
        if (f0 + 2 < half) { const f32x2 a = *reinterpret_cast<const f32x2*>(zr + col0 + 2); v[2] = a[0]; v[3] = a[1]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (f0 + j < half) ? zr[col0 + j] : 0.0f;
    }
    return v;
}
template <int HT>
__device__ __forceinline__ void store_row_half(int t, int ft, const f32x4& x, float* __restrict__ zr, int half, int g, int vw) {
    const int hh = t / HT, tt = t % HT, f0 = 32 * tt + 16 * ft + 4 * g, col0 = hh * half + f0;
    if (vw == 4) {
        if (f0 < half) *reinterpret_cast<f32x4*>(zr + col0) = x;
    } else if (vw == 2) {
        if (f0 < half) { f32x2 a = {x[0], x[1]}; *reinterpret_cast<f32x2*>(zr + col0) = a; }
        if (f0 + 2 < half) { f32x2 a = {x[2], x[3]}; *reinterpret_cast<f32x2*>(zr + col0 + 2) = a; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (f0 + j < half) zr[col0 + j] = x[j];
    }
}
// parameter-gradient dump (lsnf_layout.h LsnfDumpLayout): half-unit (nt, ft) of a (B, ld) row-major tensor in natural feature
// order, row pointer zr = base + sample * ld; v4: ld % 4 == 0 (rows 16-byte aligned)
__device__ __forceinline__ void store_plain_half(const f32x4& x, float* __restrict__ zr, int ld, int nt, int ft, int g, bool v4) {
    const int c0 = 32 * nt + 16 * ft + 4 * g;
    if (v4) {
        if (c0 < ld) *reinterpret_cast<f32x4*>(zr + c0) = x;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < ld) zr[c0 + j] = x[j];
    }
}
// sum over the 4 lane groups of a per-sample value
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

}  // namespace
