// lsnf_l16.h -- building blocks of the bf16x3 throughput kernels (lsnf_fwd3.hip, lsnf_bwd3.hip): the operand split, the
// LDS-DMA weight-panel pipeline for 6 KiB (three-term) fragments, and the "L16" lane layout on v_mfma_f32_16x16x32_bf16
// (lsnf_layout.h): a wave's 32 samples are two sample tiles st of 16; lane = (n = lane & 15, g = lane >> 4); register
// (2*ft + st)*4 + r of a 32-feature activation tile holds feature 16*ft + 4*g + r of sample 16*st + n.
#pragma once
#include "lsnf_device.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Operand split of the L16 kernels, chosen per translation unit:
//   LSNF_L16_PARTS 3 (default): three bf16 terms per operand, the six products of weight 2^-9(i+j) <= 2^-18 kept
//   LSNF_L16_PARTS 2          : two fp16 terms per operand (11 + 11 significand bits), the three products (1,1) (1,2) (2,1)
//                               kept: dropped w2*x2 <= 2^-22 |w||x|; operands must stay inside fp16's range (|x| < 65504)
#ifndef LSNF_L16_PARTS
#define LSNF_L16_PARTS 3
#endif
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#if LSNF_L16_PARTS == 3
typedef bf16x8 l16_op8;
#define LSNF_L16_TERMS(M) M(2, 0) M(0, 2) M(1, 1) M(1, 0) M(0, 1) M(0, 0)
#define LSNF_L16_NTERMS 6
#define LSNF_L16_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
typedef f16x8 l16_op8;
#define LSNF_L16_TERMS(M) M(1, 0) M(0, 1) M(0, 0)
#define LSNF_L16_NTERMS 3
#define LSNF_L16_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif
// floats of one (n-tile, k-tile) weight fragment: 2 feature halves x PARTS x 1 KiB
#define L16_FRAG_FLOATS (512 * LSNF_L16_PARTS)

// the terms of one k-step (16 features of one 32-sample tile), B-operand order
struct Split3 { l16_op8 p[LSNF_L16_PARTS]; };

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ f16x2 pk_f16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_convertvector(v, f16x2);                                     // round to nearest even
}
// LDS-DMA of KB KiB (1 KiB per wave-instruction), as lsnf_issue_panel -- but in the instruction's SGPR-base form
// (global_load_lds_dwordx4 voffset, s[base:base+1]): the wave-uniform address arithmetic stays on the scalar unit and the
// per-lane part is ONE loop-invariant register (lane * 16).  Through __builtin_amdgcn_global_load_lds hipcc forms a 64-bit
// per-lane address with a v_lshl_add_u64 per piece (the builtin's selection has no saddr pattern here), i.e. VALU work at every
// phase boundary -- exposed in the phase-separated kernels, competing for the issue port in the pipelined one.  The hardware
// counts these loads in vmcnt like any other; every consumer waits with an explicit s_waitcnt vmcnt(0) (lsnf_panel_barrier).
template <int KB, int NW>
__device__ __forceinline__ void issue_kib(const float* __restrict__ gsrc, float* lbuf, int wave, int lane) {
    constexpr int PER_WAVE = (KB + NW - 1) / NW;
    const char* base = reinterpret_cast<const char*>(gsrc);
    const unsigned lds0 = (unsigned)(size_t)((LSNF_AS3 char*)lbuf);                       // LDS byte address of the buffer
    const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll
    for (int s = 0; s < PER_WAVE; ++s) {
        const int seg = s * NW + wave;
        if (KB % NW == 0 || seg < KB) {      // wave-uniform
            const char* sb = base + (size_t)seg * 1024u;
            const unsigned m0v = lds0 + (unsigned)seg * 1024u;
            unsigned keep_m0;        // (m0 is reserved to the compiler: saved and restored around the instruction that needs it)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(lane_off), "s"(sb), "s"(m0v) : "memory");
        }
    }
}
// one 1 KiB piece (piece s of this wave: segment s * NW + wave) of the same transfer -- for kernels that spread a buffer's
// pieces over the steps of the phase before instead of issuing them in one burst behind the barrier
template <int NW>
__device__ __forceinline__ void issue_piece(const float* __restrict__ gsrc, float* lbuf, int s, int wave, int lane) {
    const int seg = s * NW + wave;
    const char* sb = reinterpret_cast<const char*>(gsrc) + (size_t)seg * 1024u;
    const unsigned m0v = (unsigned)(size_t)((LSNF_AS3 char*)lbuf) + (unsigned)seg * 1024u;
    const unsigned lane_off = (unsigned)lane * 16u;
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(lane_off), "s"(sb), "s"(m0v) : "memory");
}
template <int NW>
struct Pipe3 {
    float* buf0; int slot, cur, wave, lane;
    template <int KB> __device__ __forceinline__ void prime(const float* src) { issue_kib<KB, NW>(src, buf0, wave, lane); cur = 0; }
    template <int KB_NEXT> __device__ __forceinline__ const float* acquire(const float* next) {
        lsnf_panel_barrier();
        if (next != nullptr) issue_kib<KB_NEXT, NW>(next, buf0 + (cur ^ 1) * slot, wave, lane);
        const float* ready = buf0 + cur * slot;
        cur ^= 1;
        return ready;
    }
};
__device__ __forceinline__ constexpr int first_kib(int NT, int KT) { return 2 * LSNF_L16_PARTS * KT * (NT >= 2 ? 2 : 1); }

// the six kept terms (weight part, activation part), smallest first
#define LSNF_F3_TERMS(M) M(2, 0) M(0, 2) M(1, 1) M(1, 0) M(0, 1) M(0, 0)

typedef float f32x4v __attribute__((ext_vector_type(4)));

// feature offset (within a 32-feature tile) of the first of the 4 registers (ft, *) of lane group g
__device__ __forceinline__ constexpr int l16_feat0(int ft, int g) { return 16 * ft + 4 * g; }

// (R: long or int row indices -- 32-bit ones halve the registers that stay live from the prologue to the final stores)
template <int HT, class R>
__device__ __forceinline__ f32x16 l16_load_tile(int t, const float* __restrict__ z, const R* rows, int nz, int half, int g, int vw) {
    f32x16 x;
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int f0 = 32 * tt + l16_feat0(ft, g), col0 = hh * half + f0, b = (2 * ft + st) * 4;
            const float* zr = z + (long)rows[st] * (long)nz;
            if (vw == 4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (f0 < half) v = *reinterpret_cast<const f32x4*>(zr + col0);
                x[b] = v[0]; x[b + 1] = v[1]; x[b + 2] = v[2]; x[b + 3] = v[3];
            } else if (vw == 2) {
                f32x2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
                if (f0 < half) v0 = *reinterpret_cast<const f32x2*>(zr + col0);
                if (f0 + 2 < half) v1 = *reinterpret_cast<const f32x2*>(zr + col0 + 2);
                x[b] = v0[0]; x[b + 1] = v0[1]; x[b + 2] = v1[0]; x[b + 3] = v1[1];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[b + j] = (f0 + j < half) ? zr[col0 + j] : 0.0f;
            }
        }
    return x;
}
template <int HT, class R>
__device__ __forceinline__ void l16_store_tile(int t, const f32x16& x, float* __restrict__ z, const R* rows, const bool* live,
                                               int nz, int half, int g, int vw) {
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (!live[st]) continue;
            const int f0 = 32 * tt + l16_feat0(ft, g), col0 = hh * half + f0, b = (2 * ft + st) * 4;
            float* zr = z + (long)rows[st] * (long)nz;
            if (vw == 4) {
                if (f0 < half) { f32x4 v = {x[b], x[b + 1], x[b + 2], x[b + 3]}; *reinterpret_cast<f32x4*>(zr + col0) = v; }
            } else if (vw == 2) {
                if (f0 < half) { f32x2 v = {x[b], x[b + 1]}; *reinterpret_cast<f32x2*>(zr + col0) = v; }
                if (f0 + 2 < half) { f32x2 v = {x[b + 2], x[b + 3]}; *reinterpret_cast<f32x2*>(zr + col0 + 2) = v; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f0 + j < half) zr[col0 + j] = x[b + j];
            }
        }
}
// parameter-gradient dump (lsnf_layout.h LsnfDumpLayout): tile t of a (B, ld) row-major tensor in natural feature order;
// v4: ld % 4 == 0 (rows 16-byte aligned)
template <class R>
__device__ __forceinline__ void l16_store_plain(const f32x16& x, float* __restrict__ base, const R* rows, const bool* live,
                                                int ld, int t, int g, bool v4) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (!live[st]) continue;
            const int c0 = 32 * t + 16 * ft + 4 * g, b = (2 * ft + st) * 4;
            float* zr = base + (long)rows[st] * (long)ld;
            if (v4) {
                if (c0 < ld) { f32x4 v = {x[b], x[b + 1], x[b + 2], x[b + 3]}; *reinterpret_cast<f32x4*>(zr + c0) = v; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < ld) zr[c0 + j] = x[b + j];
            }
        }
}
// The same tile into the TILED form of a (B, ld) tensor (ld % 16 == 0): per 32-sample tile, 16-byte unit
// ((fg >> 2) * 2 + (s >> 4)) * 64 + (s & 15) * 4 + (fg & 3) for feature group fg = feature / 4 and sample s -- one store instruction
// (ft, st) writes 1 KiB of consecutive bytes (the row-major form: 16 rows x 64 bytes), and the batch contraction reads 256-byte runs
// (lsnf_params3.hip).  The tile is written whole: rows past the batch hold what the lanes hold.
__device__ __forceinline__ void l16_store_tiled(const f32x16& x, float* __restrict__ base, size_t tile32, int ld, int t, int n, int g) {
    f32x4* p = reinterpret_cast<f32x4*>(base + tile32 * 32 * (size_t)ld);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (32 * t + 16 * ft >= ld) continue;          // (a padded half of the last feature tile: past the tile's bytes)
            const int b = (2 * ft + st) * 4;
            f32x4 v = {x[b], x[b + 1], x[b + 2], x[b + 3]};
            p[(((2 * t + ft) * 2 + st) * 16 + n) * 4 + g] = v;
        }
}
// bias block of one n-tile ([h][r] order of the 32x32 layout, lsnf_prep.hip bias_feature) -> L16 accumulators
__device__ __forceinline__ f32x16 l16_bias_init(const float* cst, int g) {
    f32x16 a;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(cst + (g & 1) * 16 + 4 * (2 * ft + (g >> 1)));
#pragma unroll
        for (int st = 0; st < 2; ++st) { const int b = (2 * ft + st) * 4; a[b] = v[0]; a[b + 1] = v[1]; a[b + 2] = v[2]; a[b + 3] = v[3]; }
    }
    return a;
}
// one activation tile -> the three bf16 terms of its two sample tiles (B operands of K = 32: slots 0..3 from ft = 0, 4..7 from ft = 1)
__device__ __forceinline__ void l16_split_tile(const f32x16& x, Split3& s0, Split3& s1) {
    u32x4 w[2][LSNF_L16_PARTS];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // pair q: ft = q >> 1, registers 2*(q&1), 2*(q&1)+1
            const int b = (2 * (q >> 1) + st) * 4 + 2 * (q & 1);
            float a = x[b], c = x[b + 1];
#if LSNF_L16_PARTS == 3
            const unsigned p1 = pk_bf16(a, c);
            a -= __builtin_bit_cast(float, p1 << 16); c -= __builtin_bit_cast(float, p1 & 0xffff0000u);
            const unsigned p2 = pk_bf16(a, c);
            a -= __builtin_bit_cast(float, p2 << 16); c -= __builtin_bit_cast(float, p2 & 0xffff0000u);
            w[st][0][q] = p1; w[st][1][q] = p2; w[st][2][q] = pk_bf16(a, c);
#else
            const f16x2 h1 = pk_f16(a, c);
            {   // residuals a - x1.lo, c - x1.hi in one v_fma_mix_f32 each (f16 operand read in place: no v_cvt_f32_f16)
                const unsigned hp = __builtin_bit_cast(unsigned, h1);
                float ra, rc;
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hp), "v"(a));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rc) : "v"(hp), "v"(c));
                a = ra; c = rc;
            }
            w[st][0][q] = __builtin_bit_cast(unsigned, h1); w[st][1][q] = __builtin_bit_cast(unsigned, pk_f16(a, c));
#endif
        }
#pragma unroll
    for (int i = 0; i < LSNF_L16_PARTS; ++i) { s0.p[i] = __builtin_bit_cast(l16_op8, w[0][i]); s1.p[i] = __builtin_bit_cast(l16_op8, w[1][i]); }
}
template <int KT>
__device__ __forceinline__ void l16_split_tiles(const f32x16* x, Split3* out) {   // out[2*KT]: [kt][st]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) l16_split_tile(x[kt], out[2 * kt], out[2 * kt + 1]);
}

// NTILES n-tiles x KT k-tiles out of one LDS buffer.  A step = (n-tile, k-tile, ft): 3 fragment reads (the three weight
// parts of 16 output features) feed 12 MFMAs (6 terms x 2 sample tiles); the next step's reads are issued first.
template <int KT, int NTILES>
__device__ __forceinline__ void l16_panel_mma3(f32x16& acc0, f32x16& acc1, const Split3* in, const float* lbuf, int lane) {
    const l16_op8* wp = reinterpret_cast<const l16_op8*>(lbuf) + lane;
    constexpr int STEPS = 2 * KT * NTILES;
    __builtin_amdgcn_sched_barrier(0);
    l16_op8 a[LSNF_L16_PARTS];
#pragma unroll
    for (int p = 0; p < LSNF_L16_PARTS; ++p) a[p] = wp[p * 64];
    __builtin_amdgcn_sched_group_barrier(0x100, LSNF_L16_PARTS, 0);
    f32x4v c[2][2][2];                                   // [tile][ft][st]
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) { c[0][ft][st][r] = acc0[(2 * ft + st) * 4 + r]; c[1][ft][st][r] = acc1[(2 * ft + st) * 4 + r]; }
#pragma unroll
    for (int idx = 0; idx < STEPS; ++idx) {
        const int tile = idx / (2 * KT), kt = (idx % (2 * KT)) / 2, ft = idx & 1;
        l16_op8 na[LSNF_L16_PARTS];
#pragma unroll
        for (int p = 0; p < LSNF_L16_PARTS; ++p) na[p] = a[p];
        if (idx + 1 < STEPS) {
#pragma unroll
            for (int p = 0; p < LSNF_L16_PARTS; ++p) na[p] = wp[((idx + 1) * LSNF_L16_PARTS + p) * 64];
        }
#define LSNF_F3_MMA(WI, XI)                                                                                                        \
        c[tile][ft][0] = LSNF_L16_MFMA(a[WI], in[2 * kt + 0].p[XI], c[tile][ft][0], 0, 0, 0);         \
        c[tile][ft][1] = LSNF_L16_MFMA(a[WI], in[2 * kt + 1].p[XI], c[tile][ft][1], 0, 0, 0);
        LSNF_L16_TERMS(LSNF_F3_MMA)
#undef LSNF_F3_MMA
        if (idx + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, LSNF_L16_PARTS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * LSNF_L16_NTERMS, 0);
#pragma unroll
        for (int p = 0; p < LSNF_L16_PARTS; ++p) a[p] = na[p];
    }
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[(2 * ft + st) * 4 + r] = c[0][ft][st][r]; if (NTILES == 2) acc1[(2 * ft + st) * 4 + r] = c[1][ft][st][r]; }
}

template <int NT, int KT, int NEXT_KIB, class Pipe, class Init, class Post>
__device__ __forceinline__ void l16_gemm_stage3(Pipe& pipe, const float* gsrc, const float* gnext, f32x16* out, const Split3* in,
                                                Init&& init, Post&& post) {
    constexpr int NSP = (NT + 1) / 2;
    lsnf_static_for<NSP>([&](auto qc) {
        constexpr int q = decltype(qc)::value, t0 = 2 * q, cnt = (NT - t0 >= 2) ? 2 : 1;
        const float* lb;
        if constexpr (q + 1 < NSP) {
            constexpr int cn = (NT - (t0 + 2) >= 2) ? 2 : 1;
            lb = pipe.template acquire<2 * LSNF_L16_PARTS * KT * cn>(gsrc + (t0 + 2) * KT * L16_FRAG_FLOATS);
        } else {
            lb = pipe.template acquire<NEXT_KIB>(gnext);
        }
        out[t0] = init(t0);
        if constexpr (cnt == 2) {
            out[t0 + 1] = init(t0 + 1);
            l16_panel_mma3<KT, 2>(out[t0], out[t0 + 1], in, lb, pipe.lane);
            out[t0 + 1] = post(out[t0 + 1], t0 + 1);
        } else {
            l16_panel_mma3<KT, 1>(out[t0], out[t0], in, lb, pipe.lane);
        }
        out[t0] = post(out[t0], t0);
    });
}

// sum over the 4 lane groups of a per-sample value (lanes n, n+16, n+32, n+48)
__device__ __forceinline__ float l16_group_sum(float v) { return lsnf_pair_add32(lsnf_pair_add16(v)); }
// relu masks of one tile in the stash's (32x32-layout) word format: lanes with g < 2 end up holding the word of
// stash lane 16*st + n + 32*g for st = 0 / 1 (see the derivation in DESIGN.md section 4)
__device__ __forceinline__ void l16_store_masks(unsigned* words, const f32x16& a, int n, int g) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        unsigned c = 0;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) c |= (a[(2 * ft + st) * 4 + r] > 0.0f ? 1u : 0u) << (4 * (2 * ft + (g >> 1)) + r);
        c = lsnf_pair_or32(c);
        if (g < 2) words[16 * st + n + 32 * g] = c;
    }
}
// sigma tile of the stash ([q][lane32][4] floats, q = feature >> 3, lane32 = sample + 32*((feature >> 2) & 1))
__device__ __forceinline__ void l16_store_sigma(float* tile_base, int t, const f32x16& sg, int n, int g) {
    f32x4* p = reinterpret_cast<f32x4*>(tile_base + (size_t)t * 1024);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int b = (2 * ft + st) * 4;
            f32x4 v = {sg[b], sg[b + 1], sg[b + 2], sg[b + 3]};
            p[(2 * ft + (g >> 1)) * 64 + 16 * st + n + 32 * (g & 1)] = v;
        }
}

}  // namespace
