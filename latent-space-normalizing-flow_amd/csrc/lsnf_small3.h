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
// (no contraction: when x is a product that was formed just before -- the coupling's (v2 + t) * sigma -- "x - p1" would become
//  fma(v2 + t, sigma, -p1): the terms would then describe the UNROUNDED product, not the fp32 value that is stored as the block's
//  output and that every other kernel sees)
__device__ __forceinline__ void split4(const f32x4& x, u32x2* out /*[3]*/) {
#pragma clang fp contract(off)
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
// per-lane address of the fragments of half-unit (nt, ft) of a stage with KT k-tiles; fragment (kt, part) = ptr[(kt * 6 + part) * 64]
template <int KT>
__device__ __forceinline__ const bf16x8* unit_ptr(const float* stage, int nt, int ft, int lane) {
    return reinterpret_cast<const bf16x8*>(stage + (size_t)nt * KT * LSNF_FRAG3_FLOATS) + ft * 3 * 64 + lane;
}
// The same for ST sample tiles that share the weight fragments: acc[st] += W_unit^T in[st], in[st] = KT B-tiles at
// in_tiles + st * st_stride.  k-major, sample tiles inner: one fragment set feeds ST independent accumulator chains
// (6 * ST MFMAs per fetched fragment triple instead of 6 -- what makes a workgroup of 16 * ST rows cost the weight
// stream of a 16-row one).  NU = 1 or 2 output half-units that share the input (v1 / v2 of S1, t / p of S4): one operand
// read feeds both.  A step = one k-tile of one sample tile = 6 * NU MFMAs; every step is its own scheduling region, inside
// which sched_group_barrier pins:
//  * the three LDS operand reads of step i+1 behind the first MFMA of step i (they land under the rest of it);
//  * REFILL IN PLACE: once the last sample tile has used the fragments of k-tile kt, the SAME registers are re-loaded with
//    the same unit's fragments of the NEXT block (r0 / r1: unit_ptr of that block).  The 3 * NU loads are spread over the
//    ST steps of k-tile kt+1, one behind an MFMA each; the last k-tile's registers are handed to the NEXT stage, which
//    issues them in its k-tile-0 steps (`carry`, CN loads).  The weight stream therefore runs exactly one block ahead of
//    its use, needs no second set of registers, and its 48 loads per wave and block are spread evenly over the block's
//    MFMAs: the CU's vector-memory path takes 16 cycles per KiB, four waves share it, and a load that finds it busy holds
//    its wave -- and with it the SIMD's matrix pipe -- at issue (loads issued in bursts at the stage boundaries made the
//    launch cost stream + chains instead of max(stream, chains): 12 + 7 us per sample tile at nz = 128).
//  * EPILOGUE UNDER THE NEXT TILE: sample tile st is final after its step of the last k-tile; epi(st) -- the stage's vector
//    work on it (activation, operand split, LDS stores: about EV VALU instructions) -- is issued between the MFMAs of tile
//    st+1's last step; only the last tile's epilogue is exposed.
//  * FILL: vector work that belongs to the PREVIOUS stage (fill(st), about FV VALU instructions per sample tile) can ride
//    under the steps of this call's first k-tile -- the forward defers the coupling of block b (sigmoid, log, (v2 + t) * sigma,
//    split, LDS store: the heaviest epilogue of the block, at the stage with the fewest MFMAs behind it) to the first half of
//    block b+1's S1, whose k-tiles 0, 1 only read the v1 half of the input.
// K0..K1 = the k-tiles of this call (a stage can be cut in two calls around a barrier: the refill rule carries over).
struct NoEpi { __device__ __forceinline__ void operator()(int) const {} };
// KS: k-tiles >= KS of the input live at in_hi (k-tile kt at in_hi + (kt - KS) tiles; same sample-tile stride) -- the backward's
// g_v, whose second half is double-buffered.
template <int KT, int K0, int K1, int ST, int NU, int EV, int CN, int FV, int KS = KT, class Epi, class Carry, class Fill>
__device__ __forceinline__ void units_mma_st(f32x4* acc0, f32x4* acc1, UFrags<KT>& f0, UFrags<KT>& f1,
                                             const bf16x8* r0, const bf16x8* r1, const float* in_tiles, int st_stride, int lane,
                                             Epi&& epi, Carry&& carry, Fill&& fill, const float* in_hi = nullptr) {
    constexpr int STEPS = (K1 - K0) * ST, NM = 6 * NU;
    auto tile_at = [&](int kt, int st) { return (kt < KS ? in_tiles + kt * S3_BTILE_FLOATS : in_hi + (kt - KS) * S3_BTILE_FLOATS) + st * st_stride; };
    __builtin_amdgcn_sched_barrier(0);
    BOp b = load_btile(tile_at(K0, 0), lane);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    lsnf_static_for<STEPS>([&](auto ic) {
        constexpr int i = decltype(ic)::value, kt = K0 + i / ST, st = i % ST;
        BOp nb = b;
        if constexpr (i + 1 < STEPS) nb = load_btile(tile_at(K0 + (i + 1) / ST, (i + 1) % ST), lane);
        constexpr bool with_epi = EV > 0 && K1 == KT && kt == KT - 1 && st >= 1;
        constexpr bool with_fill = FV > 0 && kt == K0;
        if constexpr (with_epi) epi(st - 1);
        if constexpr (with_fill) fill(st);
#define S3_MMA(WI, XI) acc0[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0.w[kt][WI], b.p[XI], acc0[st], 0, 0, 0); \
                       if (NU == 2) acc1[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1.w[kt][WI], b.p[XI], acc1[st], 0, 0, 0);
        S3_TERMS(S3_MMA)
#undef S3_MMA
        // this step's share of the loads: k-tile 0 carries the previous stage's last k-tile, k-tile kt >= 1 refills k-tile kt-1
        constexpr int TOT = kt == 0 ? CN : 3 * NU, Q0 = st * TOT / ST, Q1 = (st + 1) * TOT / ST, NL = Q1 - Q0;
        lsnf_static_for<NL>([&](auto qc) {
            constexpr int q = Q0 + decltype(qc)::value;
            if constexpr (kt == 0) carry(q);
            else if constexpr (q < 3) f0.w[kt - 1][q] = r0[((kt - 1) * 6 + q) * 64];
            else f1.w[kt - 1][q - 3] = r1[((kt - 1) * 6 + q - 3) * 64];
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (i + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        constexpr int REST = NM - 1;                             // MFMAs still to place: loads first (one per MFMA), then the vector work
        lsnf_static_for<NL>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); });
        constexpr int VW = with_epi ? EV : (with_fill ? FV : 0);
        if constexpr (VW > 0 && REST - NL > 1) {
            // (the previous tile's last MFMA has to leave the pipe first: the vector work starts one MFMA later)
            constexpr int SLOTS = REST - NL - 1, PER = (VW + SLOTS - 1) / SLOTS;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            lsnf_static_for<SLOTS>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, PER, 0); });
            __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);   // the epilogue's LDS stores
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, REST - NL, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (the groups count MFMAs, they do not name them: without a hard boundary per step the
                                                //  scheduler moves one tile's MFMAs behind the next tile's, and its epilogue with them)
        b = nb;
    });
    if constexpr (EV > 0 && K1 == KT) epi(ST - 1);
}
// the refill of the last k-tile's registers of a unit, as load q (0..2) of the next stage's carry
template <int KT>
__device__ __forceinline__ void refill_last(UFrags<KT>& f, const bf16x8* r, int q) { f.w[KT - 1][q] = r[((KT - 1) * 6 + q) * 64]; }
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
// (lane exchanges on the vector ALU: lsnf_device.h lsnf_pair_add16 / lsnf_pair_add32 / lsnf_pair_or32)
__device__ __forceinline__ unsigned pair_or32(unsigned u) { return lsnf_pair_or32(u); }
// sum over the 4 lane groups of a per-sample value (lanes n, n + 16, n + 32, n + 48)
__device__ __forceinline__ float group_sum(float v) { return lsnf_pair_add32(lsnf_pair_add16(v)); }

}  // namespace
