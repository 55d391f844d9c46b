// lsnf_small3_bwd.hip -- latency backward w.r.t. z (+ fused Langevin update) on the bf16 matrix pipe, from the forward's
// activation stash: the partner of lsnf_small3_fwd.hip (workgroups of ST sample tiles of 16 rows, L16 layout, producer-side
// bf16x3 split, weights re-loaded in place one block ahead: lsnf_small3.h units_mma_st).
// Same math and ABI entry points as lsnf_small_bwd.hip's SAVED variant (replaces autograd of train.py:316-329).
//
// Per block, last to first; wave w owns the half-units g_x1[w], g_x2[w] of the running gradient (registers):
//   B4 : g_a2 = (W3s g_t + W3p g_p) * [h2 > 0]   (masks: stash)                          GTP -> GA2
//   B3 : g_a1 = (W2' g_a2) * [h1 > 0]                                                    GA2 -> GA1
//   B2 : g_v1 = g_x1 + W1' g_a1                                                          GA1 -> GV[first half]
//   B1 : g_x  = Wa [g_v1; g_v2]                                                          GV  -> registers
//   CB : s = sigma (stash); g_t = g_v2 = g_y2*s ; g_p = (1-s)(g_y2*y2 + g_l)  of the NEXT block (blk - 1): this wave's own B1
//        output and stash slice, so it runs as B1's epilogue (under the next sample tile's MFMAs) -> GTP, GV[second half, other buffer]
// Four barriers per block.  The transposed matrices come as bf16x3 panels in the 16x16x32 operand order (off_b3b_panels).
#include <stdlib.h>
#include "lsnf_small3.h"

namespace {

template <int HT_, int WT_>
struct Small3BwdCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = LSNF_FRAG3_FLOATS;
    static constexpr int OFFB4 = 0;
    static constexpr int OFFB3 = OFFB4 + F * WT_ * 2 * HT_;
    static constexpr int OFFB2 = OFFB3 + F * WT_ * WT_;
    static constexpr int OFFB1 = OFFB2 + F * HT_ * WT_;
    static constexpr int BLOCKB = OFFB1 + F * S::NZT * S::NZT;
    static constexpr int NU2 = (2 * WT_ + 3) / 4;
};
// LDS map (floats) for ST sample tiles: GTP (g_t | g_p: 2HT B-tiles per sample tile; g_a1 reuses its front once B4 has read it),
// GA2 (WT), GV first half (HT), GV second half (HT, double-buffered: the next block's coupling backward writes it while B1 reads)
template <class C, int ST>
struct Small3BwdLds {
    static constexpr int TP = 2 * C::HT * S3_BTILE_FLOATS, HL = C::WT * S3_BTILE_FLOATS, GH = C::HT * S3_BTILE_FLOATS;
    static_assert(C::WT <= 2 * C::HT, "g_a1 is kept in the front of the g_t | g_p tiles");
    static constexpr int L_GTP = 0;
    static constexpr int L_GA2 = L_GTP + ST * TP;
    static constexpr int L_GVA = L_GA2 + ST * HL;
    static constexpr int L_GVB = L_GVA + ST * GH;
    static constexpr int L_RED = L_GVB + 2 * ST * GH;
    static constexpr int L_END = L_RED + ST * 4 * 16 * 2;
};

struct Small3BwdArgs {
    const float* panels;                 // b3b region, block 0
    const float* z_out; const float* z_saved; const float* act_saved; const float* g_z1; const float* g_logdet;
    float* g_z_in;
    const float* z_cur; const float* grad_g; const float* noise; float* z_new; float* gf_norm; float* gg_norm;
    float step, ll_scale;
    LsnfRngArgs rng;
    int ll_mode, B, nz, half, depth, vec4;
    float* dump; float* gl_total; int width;      // DUMP variant (parameter gradients, lsnf_params.hip): per block g_v, g_a1, g_a2,
                                                  // g_t, g_p written for the batch contraction; G = sum_b dL/dlogdet_b
};

__device__ __forceinline__ f32x4 mask4(f32x4 a, unsigned nib) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = ((nib >> r) & 1u) ? a[r] : 0.0f;
    return a;
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

template <class C, int ST, bool DUMP>
__global__ __launch_bounds__(256, 1) void lsnf_small3_bwd_kernel(const Small3BwdArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2, LASTU = NU2 - 1;
    using L = Small3BwdLds<C, ST>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* GTP = smem + L::L_GTP;
    float* GA2 = smem + L::L_GA2;
    float* GA1 = GTP;                                                     // (see Small3BwdLds)
    float* GVA = smem + L::L_GVA;
    float* GVB = smem + L::L_GVB;
    float* RED = smem + L::L_RED;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int vec4 = a.vec4;

    // a wave without a unit of its own (HT = 1: waves 2, 3) computes unit 0 again and stores the same values to the same LDS
    // words: no branch in the stages (global stores stay under has1 / hasw)
    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }

    const int last = a.depth - 1;
    const float* gb_last = a.panels + (size_t)last * C::BLOCKB;
    long sample[ST]; bool live[ST]; long row[ST];
    size_t wtile[ST]; int lane32[ST]; bool tile_ok[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        const size_t q = (size_t)blockIdx.x * ST + st;                    // 16-row tile q = half (q & 1) of the 32-sample stash tile q >> 1
        sample[st] = (long)q * S3_SAMPLES + n;
        live[st] = sample[st] < a.B;
        row[st] = live[st] ? sample[st] : (long)a.B - 1;
        tile_ok[st] = (long)q * S3_SAMPLES < (long)a.B;
        wtile[st] = tile_ok[st] ? (q >> 1) : 0;                           // (a tile past the batch reads stash tile 0: finite values, never stored)
        lane32[st] = 16 * (int)(q & 1) + n + 32 * (g & 1);
    }
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const int nibsh_base = 4 * (g >> 1);                                   // + 8*ft: bit offset of this lane's nibble in a mask word

    // a block's stash slice and block output (this wave's second-half unit), fetched one block ahead of their use
    f32x4 y2[ST], sg[ST]; unsigned m1[NU2][ST], m2[NU2][ST];
    auto fetch_block_state = [&](int blk) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const float* act = a.act_saved + (size_t)blk * al.per_block + wtile[st] * al.per_tile;
            const float* ysrc = (blk == a.depth - 1) ? a.z_out + row[st] * (long)a.nz : a.z_saved + ((size_t)blk * a.B + row[st]) * a.nz;
            y2[st] = load_row_half<HT>(HT + nt1, ft1, ysrc, a.half, g, vec4);
            sg[st] = reinterpret_cast<const f32x4*>(act + (size_t)nt1 * 1024)[(2 * ft1 + (g >> 1)) * 64 + lane32[st]];
            const unsigned* words = reinterpret_cast<const unsigned*>(act + al.mask_off);
#pragma unroll
            for (int i = 0; i < NU2; ++i) {
                const int nt = hw[i] >> 1, ft = hw[i] & 1;
                m1[i][st] = (words[nt * 64 + lane32[st]] >> (nibsh_base + 8 * ft)) & 0xFu;
                m2[i][st] = (words[(WT + nt) * 64 + lane32[st]] >> (nibsh_base + 8 * ft)) & 0xFu;
            }
        }
    };
    // what the gradient needs first goes out first (vmcnt completes in order): the last block's stash slice and the stack's
    // output, then the weights in order of use
    fetch_block_state(last);
    float gl[ST];
    f32x4 gx1[ST], gx2[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        const float* zo = a.z_out + row[st] * (long)a.nz;
        if (a.ll_mode) {            // L = ll_scale * sum ll: dL/dz1 = -ll_scale * z1, dL/dlogdet = ll_scale (train.py:317-320)
            gl[st] = a.ll_scale;
            const f32x4 y1 = load_row_half<HT>(nt1, ft1, zo, a.half, g, vec4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { gx1[st][r] = -a.ll_scale * y1[r]; gx2[st][r] = -a.ll_scale * y2[st][r]; }
        } else {
            gl[st] = a.g_logdet ? a.g_logdet[row[st]] : 0.0f;
            if (a.g_z1) {
                gx1[st] = load_row_half<HT>(nt1, ft1, a.g_z1 + row[st] * (long)a.nz, a.half, g, vec4);
                gx2[st] = load_row_half<HT>(HT + nt1, ft1, a.g_z1 + row[st] * (long)a.nz, a.half, g, vec4);
            } else { gx1[st] = zero4(); gx2[st] = zero4(); }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    UFrags<2 * HT> wb4[NU2];
    UFrags<WT> wb3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) wb4[i] = fetch_unit<2 * HT>(gb_last + C::OFFB4, hw[i] >> 1, hw[i] & 1, lane);
#pragma unroll
    for (int i = 0; i < NU2; ++i) wb3[i] = fetch_unit<WT>(gb_last + C::OFFB3, hw[i] >> 1, hw[i] & 1, lane);
    UFrags<WT> wb2 = fetch_unit<WT>(gb_last + C::OFFB2, nt1, ft1, lane);
    UFrags<NZT> wb1a = fetch_unit<NZT>(gb_last + C::OFFB1, nt1, ft1, lane);
    UFrags<NZT> wb1b = fetch_unit<NZT>(gb_last + C::OFFB1, HT + nt1, ft1, lane);
    __builtin_amdgcn_sched_barrier(0);

    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const bool w4 = (a.width & 3) == 0, h4 = (a.half & 3) == 0;
    if constexpr (DUMP) {   // G = sum_b dL/dlogdet_b: one atomic per workgroup
        if (wave == 0) {
            float t = 0.0f;
#pragma unroll
            for (int st = 0; st < ST; ++st) t += (live[st] && g == 0) ? gl[st] : 0.0f;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) atomicAdd(a.gl_total, t);
        }
    }
    // CB of block `blk` for sample tile st: from this wave's g_x2 (the gradient on the block's output, second half), the block's
    // sigma / y2 (fetch_block_state(blk) must have run).  gvb: the g_v second-half buffer the block's B1 will read.
    auto coupling_backward = [&](int st, int blk, float* gvb, bool real) {
        f32x4 gt, gp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gt[r] = gx2[st][r] * sg[st][r];
            gp[r] = (1.0f - sg[st][r]) * (gx2[st][r] * y2[st][r] + gl[st]);
        }
        store_half(GTP + st * L::TP + nt1 * S3_BTILE_FLOATS, ft1, gt, lane);
        store_half(GTP + st * L::TP + (HT + nt1) * S3_BTILE_FLOATS, ft1, gp, lane);
        store_half(gvb + st * L::GH + nt1 * S3_BTILE_FLOATS, ft1, gt, lane);                 // g_v2 = g_t
        if constexpr (DUMP) {
            if (real && has1 && live[st]) {
                float* dmp = a.dump + (size_t)blk * dl.per_block;
                store_plain_half(gt, dmp + dl.off_gt + sample[st] * (long)a.half, a.half, nt1, ft1, g, h4);
                store_plain_half(gp, dmp + dl.off_gp + sample[st] * (long)a.half, a.half, nt1, ft1, g, h4);
                store_row_half<HT>(HT + nt1, ft1, gt, dmp + dl.off_gv + sample[st] * (long)a.nz, a.half, g, vec4);
            }
        }
    };
#pragma unroll
    for (int st = 0; st < ST; ++st) coupling_backward(st, last, GVB, true);     // the last block's CB (buffer 0)
    __syncthreads();

    for (int blk = last; blk >= 0; --blk) {
        const float* gb = a.panels + (size_t)blk * C::BLOCKB;
        const int nb = blk > 0 ? blk - 1 : 0;                              // block 0 re-fetches its own panels: no loads under a branch
        const float* gbn = a.panels + (size_t)nb * C::BLOCKB;
        float* gvb_cur = GVB + ((last - blk) & 1) * ST * L::GH;
        float* gvb_nxt = GVB + ((last - blk + 1) & 1) * ST * L::GH;
        float* dmp = DUMP ? a.dump + (size_t)blk * dl.per_block : nullptr;

        // ---- B4: g_a2 = ([W3s W3p][g_t; g_p]) gated by h2 > 0 ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 ga[ST];
#pragma unroll
            for (int st = 0; st < ST; ++st) ga[st] = zero4();
            auto epi = [&](int st) {
                ga[st] = mask4(ga[st], m2[i][st]);
                store_half(GA2 + st * L::HL + nt * S3_BTILE_FLOATS, ft, ga[st], lane);
            };
            const bf16x8* rf = unit_ptr<2 * HT>(gbn + C::OFFB4, nt, ft, lane);
            if (i == 0) {        // carry: the last k-tile of B1's two units, THIS block's fragments (B1 runs last in the block)
                const bf16x8* c1a = unit_ptr<NZT>(gb + C::OFFB1, nt1, ft1, lane);
                const bf16x8* c1b = unit_ptr<NZT>(gb + C::OFFB1, HT + nt1, ft1, lane);
                units_mma_st<2 * HT, 0, 2 * HT, ST, 1, 26, 6, 0>(ga, ga, wb4[i], wb4[i], rf, nullptr, GTP, L::TP, lane, epi,
                    [&](int q) { if (q < 3) refill_last<NZT>(wb1a, c1a, q); else refill_last<NZT>(wb1b, c1b, q - 3); }, [](int) {});
            } else {
                const bf16x8* cp = unit_ptr<2 * HT>(gbn + C::OFFB4, hw[i > 0 ? i - 1 : 0] >> 1, hw[i > 0 ? i - 1 : 0] & 1, lane);
                units_mma_st<2 * HT, 0, 2 * HT, ST, 1, 26, 3, 0>(ga, ga, wb4[i], wb4[i], rf, nullptr, GTP, L::TP, lane, epi,
                    [&](int q) { refill_last<2 * HT>(wb4[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
            if constexpr (DUMP) {
                if (hasw[i]) {
#pragma unroll
                    for (int st = 0; st < ST; ++st)
                        if (live[st]) store_plain_half(ga[st], dmp + dl.off_ga2 + sample[st] * (long)a.width, a.width, nt, ft, g, w4);
                }
            }
        }
        __syncthreads();
        // ---- B3: g_a1 = (W2' g_a2) gated by h1 > 0 ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 ga[ST];
#pragma unroll
            for (int st = 0; st < ST; ++st) ga[st] = zero4();
            auto epi = [&](int st) {
                ga[st] = mask4(ga[st], m1[i][st]);
                store_half(GA1 + st * L::TP + nt * S3_BTILE_FLOATS, ft, ga[st], lane);
            };
            const bf16x8* rf = unit_ptr<WT>(gbn + C::OFFB3, nt, ft, lane);
            if (i == 0) {        // carry: the last k-tile of B4's last unit (next block's fragments)
                const bf16x8* cp = unit_ptr<2 * HT>(gbn + C::OFFB4, hw[LASTU] >> 1, hw[LASTU] & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 26, 3, 0>(ga, ga, wb3[i], wb3[i], rf, nullptr, GA2, L::HL, lane, epi,
                    [&](int q) { refill_last<2 * HT>(wb4[LASTU], cp, q); }, [](int) {});
            } else {
                const bf16x8* cp = unit_ptr<WT>(gbn + C::OFFB3, hw[i > 0 ? i - 1 : 0] >> 1, hw[i > 0 ? i - 1 : 0] & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 26, 3, 0>(ga, ga, wb3[i], wb3[i], rf, nullptr, GA2, L::HL, lane, epi,
                    [&](int q) { refill_last<WT>(wb3[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
            if constexpr (DUMP) {
                if (hasw[i]) {
#pragma unroll
                    for (int st = 0; st < ST; ++st)
                        if (live[st]) store_plain_half(ga[st], dmp + dl.off_ga1 + sample[st] * (long)a.width, a.width, nt, ft, g, w4);
                }
            }
        }
        __syncthreads();
        // ---- B2: g_v1 = g_x1 (direct) + W1' g_a1.  The next block's stash slice / output rows are requested here ----
        fetch_block_state(nb);                                             // (this block's are consumed: the last CB ran before B4)
        {
            f32x4 gv1[ST];
#pragma unroll
            for (int st = 0; st < ST; ++st) gv1[st] = gx1[st];
            const bf16x8* cp = unit_ptr<WT>(gbn + C::OFFB3, hw[LASTU] >> 1, hw[LASTU] & 1, lane);
            units_mma_st<WT, 0, WT, ST, 1, 24, 3, 0>(gv1, gv1, wb2, wb2, unit_ptr<WT>(gbn + C::OFFB2, nt1, ft1, lane), nullptr, GA1, L::TP, lane,
                [&](int st) { store_half(GVA + st * L::GH + nt1 * S3_BTILE_FLOATS, ft1, gv1[st], lane); },
                [&](int q) { refill_last<WT>(wb3[LASTU], cp, q); }, [](int) {});
            if constexpr (DUMP) {
                if (has1) {
#pragma unroll
                    for (int st = 0; st < ST; ++st)
                        if (live[st]) store_row_half<HT>(nt1, ft1, gv1[st], dmp + dl.off_gv + sample[st] * (long)a.nz, a.half, g, vec4);
                }
            }
        }
        __syncthreads();
        // ---- B1: g_x = Wa [g_v1; g_v2]; under its last steps: the coupling backward of the next block ----
        {
#pragma unroll
            for (int st = 0; st < ST; ++st) { gx1[st] = zero4(); gx2[st] = zero4(); }
            const bf16x8* cp = unit_ptr<WT>(gbn + C::OFFB2, nt1, ft1, lane);
            units_mma_st<NZT, 0, NZT, ST, 2, 30, 3, 0, HT>(gx1, gx2, wb1a, wb1b, unit_ptr<NZT>(gbn + C::OFFB1, nt1, ft1, lane),
                unit_ptr<NZT>(gbn + C::OFFB1, HT + nt1, ft1, lane), GVA, L::GH, lane,
                [&](int st) { coupling_backward(st, nb, gvb_nxt, blk > 0); },   // (after block 0: LDS writes nobody reads, no dump)
                [&](int q) { refill_last<WT>(wb2, cp, q); }, [](int) {}, gvb_cur);
        }
        __syncthreads();
    }

    // ---- outputs: g_z_in and / or the fused Langevin update (train.py:324-329) ----
    LsnfRngState rs = {0u, 0u, 0u, 0u, 0};
    if (a.z_new && !a.noise && a.rng.enabled) {
        const unsigned long long off = a.rng.offset + (a.rng.offset_dev ? *a.rng.offset_dev : 0ull);
        rs = {(unsigned)a.rng.seed, (unsigned)(a.rng.seed >> 32), (unsigned)off, (unsigned)(off >> 32), 1};
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        float gf2 = 0.0f, gg2 = 0.0f;
        if (has1) {
            if (live[st] && a.g_z_in) {
                float* gr = a.g_z_in + sample[st] * (long)a.nz;
                store_row_half<HT>(nt1, ft1, gx1[st], gr, a.half, g, vec4);
                store_row_half<HT>(HT + nt1, ft1, gx2[st], gr, a.half, g, vec4);
            }
            if (a.z_new) {
                const float coef = 0.5f * a.step * a.step;
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {                               // this wave's first-half and second-half unit
                    const int t = hs * HT + nt1;
                    const f32x4 gq = hs ? gx2[st] : gx1[st];
                    const f32x4 zc = load_row_half<HT>(t, ft1, a.z_cur + row[st] * (long)a.nz, a.half, g, vec4);
                    f32x4 gs = gq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) gf2 += gq[r] * gq[r];
                    if (a.grad_g) {
                        const f32x4 gg = load_row_half<HT>(t, ft1, a.grad_g + row[st] * (long)a.nz, a.half, g, vec4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { gg2 += gg[r] * gg[r]; gs[r] = gg[r] + gq[r]; }   // z_grad_g + z_grad_f (train.py:324)
                    }
                    f32x4 zn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) zn[r] = zc[r] - coef * gs[r];
                    if (a.noise) {
                        const f32x4 nv = load_row_half<HT>(t, ft1, a.noise + row[st] * (long)a.nz, a.half, g, vec4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) zn[r] = zn[r] + a.step * nv[r];                    // train.py:326
                    } else if (rs.on) {     // the same draws as every other kernel: a function of (seed, offset, global row, column)
                        const unsigned long long grow = (unsigned long long)(a.rng.row0 + sample[st]);
                        const int f0 = 32 * nt1 + 16 * ft1 + 4 * g;
                        unsigned c0 = ((unsigned)hs << 16) | (unsigned)(f0 >> 2), c1 = (unsigned)grow, c2 = rs.c2, c3 = rs.c3hi ^ (unsigned)(grow >> 32);
                        lsnf_philox4x32_10(c0, c1, c2, c3, rs.k0, rs.k1);
                        float n0, n1, n2, n3;
                        lsnf_box_muller(c0, c1, n0, n1);
                        lsnf_box_muller(c2, c3, n2, n3);
                        zn[0] += a.step * n0; zn[1] += a.step * n1; zn[2] += a.step * n2; zn[3] += a.step * n3;
                    }
                    if (live[st]) store_row_half<HT>(t, ft1, zn, a.z_new + sample[st] * (long)a.nz, a.half, g, vec4);
                }
            }
        }
        if (a.z_new && (a.gf_norm || a.gg_norm)) {      // kernel-uniform: per-sample norms of train.py:328-329
            gf2 = group_sum(gf2); gg2 = group_sum(gg2);
            if (g == 0) { RED[((st * 4 + wave) * 16 + n) * 2] = gf2; RED[((st * 4 + wave) * 16 + n) * 2 + 1] = gg2; }
        }
    }
    if (a.z_new && (a.gf_norm || a.gg_norm)) {
        __syncthreads();
        if (wave == 0 && g == 0) {
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s1 += RED[((st * 4 + w) * 16 + n) * 2]; s2 += RED[((st * 4 + w) * 16 + n) * 2 + 1]; }
                if (live[st]) {
                    if (a.gf_norm) a.gf_norm[sample[st]] = sqrtf(s1);
                    if (a.gg_norm) a.gg_norm[sample[st]] = sqrtf(s2);
                }
            }
        }
    }
}

template <class C, int ST>
hipError_t launch_small3_bwd_st(const Small3BwdArgs& a, hipStream_t stream) {
    if constexpr ((size_t)Small3BwdLds<C, ST>::L_END * sizeof(float) > 160 * 1024) {
        return hipErrorInvalidValue;                 // (this shape does not fit: not instantiated)
    } else {
        const size_t lds = (size_t)Small3BwdLds<C, ST>::L_END * sizeof(float);
        auto kern = a.dump ? lsnf_small3_bwd_kernel<C, ST, true> : lsnf_small3_bwd_kernel<C, ST, false>;
        static unsigned long long lds_ok[2] = {0, 0};
        if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok[a.dump ? 1 : 0]); e != hipSuccess) return e;
        const unsigned grid = (unsigned)((a.B + ST * S3_SAMPLES - 1) / (ST * S3_SAMPLES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
        return hipGetLastError();
    }
}
// rows per workgroup by batch size, as the forward (lsnf_small3_fwd.hip launch_small3_fwd); LSNF_SMALL3_ST forces a shape
template <class C>
hipError_t launch_small3_bwd(const Small3BwdArgs& a, hipStream_t stream) {
    static const char* env = getenv("LSNF_SMALL3_ST");
    const int st = env ? atoi(env) : (a.B <= 256 * 16 ? 1 : (a.B <= 256 * 32 ? 2 : 4));
    hipError_t e = hipErrorInvalidValue;
    if (st >= 4) e = launch_small3_bwd_st<C, 4>(a, stream);
    if (e == hipErrorInvalidValue && st >= 2) e = launch_small3_bwd_st<C, 2>(a, stream);
    if (e == hipErrorInvalidValue) e = launch_small3_bwd_st<C, 1>(a, stream);
    return e;
}
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); needs the activation stash; hipErrorInvalidValue = not covered
hipError_t lsnf_launch_small3_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                         const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                                         float ll_scale, float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                         float* dump, float* gl_total) {
    if (!act_saved) return hipErrorInvalidValue;
    Small3BwdArgs a;
    a.dump = dump; a.gl_total = gl_total; a.width = g.width;
    a.panels = plan + g.off_b3b_panels;
    a.z_out = z_out; a.z_saved = z_saved; a.act_saved = act_saved; a.g_z1 = g_z1; a.g_logdet = g_logdet; a.g_z_in = g_z_in;
    a.z_cur = nullptr; a.grad_g = nullptr; a.noise = nullptr; a.z_new = nullptr; a.gf_norm = nullptr; a.gg_norm = nullptr; a.step = 0.f;
    a.rng = LsnfRngArgs{0ull, 0ull, nullptr, 0ll, 0};
    if (lv) { a.z_cur = lv->z_cur; a.grad_g = lv->grad_g; a.noise = lv->noise; a.z_new = lv->z_new; a.gf_norm = lv->gf_norm;
              a.gg_norm = lv->gg_norm; a.step = lv->step; a.rng = lv->rng; }
    a.ll_scale = ll_scale; a.ll_mode = ll_mode; a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small3_bwd<Small3BwdCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small3_bwd<Small3BwdCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small3_bwd<Small3BwdCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
