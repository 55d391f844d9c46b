// lsnf_bwd.hip -- fused backward of the whole stack w.r.t. z, one launch.
// Replaces what autograd does for train.py:316-323 (z_grad_f = d(-sum ll)/dz through
// model.py:391-422 x f_depth): per block, last to first, with the running gradient
// g = dL/d[z1|z2'] in VGPRs:
//   recompute  h1,h2 (relu masks), shift t, pre-sigmoid p from the SAVED block output (z1 = v1)
//   g_v2 = g_y2*s ; g_t = g_v2 ; g_p = (1-s)*(g_y2*y2 + g_l)      [s = sigmoid(p), y2 = (v2+t)*s]
//   g_h2 = W3s g_t + W3p g_p ; g_a2 = g_h2 * [h2>0] ; g_h1 = W2' g_a2 ; g_a1 = g_h1 * [h1>0]
//   g_v1 = g_v1 + W1' g_a1 ;  g_x = Wa [g_v1; g_v2]               (actnorm folded into Wa)
// dL/dlogdet (g_l) is constant through the stack because logdet_out = logdet_in + (...).
// Activation memory: only the block outputs (z_saved from lsnf_forward) -- the MLP is recomputed.
#include "lsnf_device.h"

namespace {

template <int HT_, int WT_>
struct BwdCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    using S::HT; using S::WT; using S::NZT;
    static constexpr int CONST_USED = 32 * (S::P2 + S::P3 + S::P4);   // only the MLP biases are needed: S2, S3, S4 bias blocks
};

struct BwdArgs {
    const float* fwd_consts; const float* fwd_panels; const float* bwd_panels;
    const float* z_out; const float* z_saved; const float* g_z1; const float* g_logdet;
    float* g_z_in;          // may be NULL when only parameter gradients are wanted
    float* dump;            // DUMP variant: per-block intermediates for the parameter gradients (lsnf_layout.h)
    float* gl_total;        // DUMP variant: += sum_b dL/dlogdet_b
    const float* act_saved; // SAVED variant: activation stash of the forward (sigma, relu masks): no MLP recompute
    // fused Langevin update (train.py:324-329): z_new = z_cur - 0.5 s^2 (grad_g + g_z_in) + s * noise
    const float* z_cur; const float* grad_g; const float* noise; float* z_new; float* gf_norm; float* gg_norm;
    float step;
    LsnfRngArgs rng;        // Langevin update: in-kernel noise when `noise` is NULL and rng.enabled
    float ll_scale;
    int ll_mode, B, nz, half, width, depth, vec4;
};

// plain-pad tiles (feature f = 32*t + o(r,h), valid f < ncols) -> dense (B, ncols) row-major
template <int T>
__device__ __forceinline__ void store_plain(const f32x16* x, float* __restrict__ dst, long row, int ncols, int h) {
    float* d = dst + row * (long)ncols;
    const bool v4 = (ncols & 3) == 0;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * t + 8 * g + 4 * h;
            if (v4) {
                if (f0 < ncols) {
                    f32x4 v = {x[t][4 * g + 0], x[t][4 * g + 1], x[t][4 * g + 2], x[t][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(d + f0) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f0 + j < ncols) d[f0 + j] = x[t][4 * g + j];
            }
        }
}

// NW waves per workgroup (4: two workgroups per CU, 8: one; see lsnf_fwd.hip)
// Wide MLPs (WT = 4, e.g. f_width = 128) need ~300 live VGPRs in the backward: give them the whole register file
// (one wave per SIMD) instead of spilling 80-110 registers to scratch at two waves per SIMD.
// SAVED: sigma and the relu masks come from the forward's activation stash (lsnf_forward act_saved) instead of being
// recomputed: the three MLP GEMMs (1/3 of this kernel's MFMA work) disappear.  DUMP needs the h1/h2 VALUES, so it
// always recomputes (DUMP implies !SAVED).
template <class C, bool DUMP, int NW, bool SAVED>
__global__ __launch_bounds__(64 * NW, ((C::WT >= 4 && !SAVED) ? 1 : 2)) void lsnf_bwd_z_kernel(const BwdArgs a) {
    static_assert(!(DUMP && SAVED), "the parameter-gradient dump needs the recomputed activations");
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                   // depth * CONST_USED
    const int tid = threadIdx.x;
    LsnfPipeT<NW> pipe;
    pipe.buf0 = smem + a.depth * C::CONST_USED;
    pipe.slot = C::SLOT;
    pipe.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pipe.lane = tid & 63;
    const int lane = pipe.lane, m = lane & 31, h = lane >> 5;

    if constexpr (SAVED) pipe.template prime<2 * HT>(a.bwd_panels + (size_t)(a.depth - 1) * C::BWD_BLOCK + C::OFF_B4);
    else pipe.template prime<HT>(a.fwd_panels + (size_t)(a.depth - 1) * C::FWD_BLOCK + C::OFF_S2);
    for (int i = tid; i < a.depth * C::CONST_USED; i += 64 * NW) {
        const int blk = i / C::CONST_USED, r = i % C::CONST_USED;
        cst[i] = a.fwd_consts[blk * C::FWD_CONST + 32 * C::P1 + r];
    }
    const long sample = ((long)blockIdx.x * NW + pipe.wave) * 32 + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    const int vec4 = a.vec4;

    // upstream gradient on the stack output
    f32x16 gx[NZT];
    float gl;
    if (a.ll_mode) {       // L = ll_scale * sum ll : dL/dz1 = -ll_scale * z1, dL/dlogdet = ll_scale (train.py:317-320)
        lsnf_load_rows<HT>(gx, a.z_out, row, a.nz, a.half, h, vec4);
#pragma unroll
        for (int t = 0; t < NZT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[t][r] = -a.ll_scale * gx[t][r];
        gl = a.ll_scale;
    } else {
        if (a.g_z1) lsnf_load_rows<HT>(gx, a.g_z1, row, a.nz, a.half, h, vec4);
        else {
#pragma unroll
            for (int t = 0; t < NZT; ++t) gx[t] = lsnf_zero16();
        }
        gl = a.g_logdet ? a.g_logdet[row] : 0.0f;
    }
    if constexpr (DUMP) {   // G = sum_b dL/dlogdet_b : one atomic per wave
        float t = (live && h == 0) ? gl : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) atomicAdd(a.gl_total, t);
    }
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    size_t wtile = (size_t)blockIdx.x * NW + pipe.wave;              // this wave's 32-sample tile (clamped: waves past
    if (wtile * 32 >= (size_t)a.B) wtile = (size_t)(a.B - 1) / 32;   // the batch read a valid tile and store nothing)

    for (int blk = a.depth - 1; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_USED;
        const float* gf = a.fwd_panels + (size_t)blk * C::FWD_BLOCK;
        const float* gb = a.bwd_panels + (size_t)blk * C::BWD_BLOCK;
        const float* gnext = blk == 0 ? nullptr
                           : (SAVED ? a.bwd_panels + (size_t)(blk - 1) * C::BWD_BLOCK + C::OFF_B4
                                    : a.fwd_panels + (size_t)(blk - 1) * C::FWD_BLOCK + C::OFF_S2);
        const float* ysrc = (blk == a.depth - 1) ? a.z_out : a.z_saved + (size_t)blk * a.B * a.nz;

        f32x16 y[NZT];     // block output [v1 | y2]
        lsnf_load_rows<HT>(y, ysrc, row, a.nz, a.half, h, vec4);
        unsigned m1[WT], m2[WT];
        f32x16 tp[2 * HT];
        float* dmp = nullptr;
        if constexpr (SAVED) {
            const float* act = a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile;
#pragma unroll
            for (int t = 0; t < WT; ++t) { m1[t] = *lsnf_act_mask_ptr(act, al.mask_off, t, lane); m2[t] = *lsnf_act_mask_ptr(act, al.mask_off, WT + t, lane); }
            // ---- coupling backward with the stashed sigma: tp[0..HT) <- g_t (= g_v2), tp[HT..2HT) <- g_p ----
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const f32x16 sg = lsnf_act_load_sigma(act, t, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float gy2 = gx[HT + t][r];
                    tp[t][r] = gy2 * sg[r];
                    tp[HT + t][r] = (1.0f - sg[r]) * (gy2 * y[HT + t][r] + gl);
                }
            }
        } else {
        // ---- recompute the MLP (forward panels S2..S4) ----
        f32x16 h1[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* nxt = (nt + 1 < WT) ? gf + C::OFF_S2 + (nt + 1) * HT * LSNF_FRAG_FLOATS : gf + C::OFF_S3;
            const float* lb = (nt + 1 < WT) ? pipe.template acquire<HT>(nxt) : pipe.template acquire<WT>(nxt);
            h1[nt] = lsnf_bias_init(cb + 32 * nt, h);
            lsnf_panel_mma<HT>(h1[nt], y, lb, lane);
            h1[nt] = lsnf_relu16(h1[nt]);
        });
        f32x16 h2[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* nxt = (nt + 1 < WT) ? gf + C::OFF_S3 + (nt + 1) * WT * LSNF_FRAG_FLOATS : gf + C::OFF_S4;
            const float* lb = pipe.template acquire<WT>(nxt);
            h2[nt] = lsnf_bias_init(cb + 32 * (C::P2 + nt), h);
            lsnf_panel_mma<WT>(h2[nt], h1, lb, lane);
            h2[nt] = lsnf_relu16(h2[nt]);
        });
#pragma unroll
        for (int t = 0; t < WT; ++t) m1[t] = lsnf_posmask16(h1[t]);
        lsnf_static_for<2 * HT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < 2 * HT) lb = pipe.template acquire<WT>(gf + C::OFF_S4 + (nt + 1) * WT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<2 * HT>(gb + C::OFF_B4);
            tp[nt] = lsnf_bias_init(cb + 32 * (C::P2 + C::P3 + nt), h);
            lsnf_panel_mma<WT>(tp[nt], h2, lb, lane);
        });
#pragma unroll
        for (int t = 0; t < WT; ++t) m2[t] = lsnf_posmask16(h2[t]);
        if constexpr (DUMP) {
            dmp = a.dump + (size_t)blk * dl.per_block;
            if (live) {
                store_plain<WT>(h1, dmp + dl.off_h1, sample, a.width, h);
                store_plain<WT>(h2, dmp + dl.off_h2, sample, a.width, h);
            }
        }

        // ---- coupling backward: tp[0..HT) <- g_t (= g_v2), tp[HT..2HT) <- g_p ----
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(tp[HT + t][r], sig, lsig);
                const float gy2 = gx[HT + t][r];
                tp[t][r] = gy2 * sig;
                tp[HT + t][r] = (1.0f - sig) * (gy2 * y[HT + t][r] + gl);
            }
        }   // !SAVED
        if constexpr (DUMP) {
            if (live) {
                store_plain<HT>(tp, dmp + dl.off_gt, sample, a.half, h);
                store_plain<HT>(tp + HT, dmp + dl.off_gp, sample, a.half, h);
            }
        }
        // ---- B4: g_h2 = [W3s W3p] [g_t; g_p], relu mask ----
        f32x16 gh2[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < WT) lb = pipe.template acquire<2 * HT>(gb + C::OFF_B4 + (nt + 1) * 2 * HT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<WT>(gb + C::OFF_B3);
            gh2[nt] = lsnf_zero16();
            lsnf_panel_mma<2 * HT>(gh2[nt], tp, lb, lane);
            gh2[nt] = lsnf_apply_mask16(gh2[nt], m2[nt]);
        });
        if constexpr (DUMP) { if (live) store_plain<WT>(gh2, dmp + dl.off_ga2, sample, a.width, h); }
        // ---- B3: g_h1 = W2' g_a2, relu mask ----
        f32x16 gh1[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* lb = pipe.template acquire<WT>((nt + 1 < WT) ? gb + C::OFF_B3 + (nt + 1) * WT * LSNF_FRAG_FLOATS : gb + C::OFF_B2);
            gh1[nt] = lsnf_zero16();
            lsnf_panel_mma<WT>(gh1[nt], gh2, lb, lane);
            gh1[nt] = lsnf_apply_mask16(gh1[nt], m1[nt]);
        });
        if constexpr (DUMP) { if (live) store_plain<WT>(gh1, dmp + dl.off_ga1, sample, a.width, h); }
        // ---- B2: g_v1 = g_v1(direct) + W1' g_a1 ;  gv = [g_v1 ; g_v2] ----
        f32x16 gv[NZT];
        lsnf_static_for<HT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < HT) lb = pipe.template acquire<WT>(gb + C::OFF_B2 + (nt + 1) * WT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<NZT>(gb + C::OFF_B1);
            gv[nt] = gx[nt];
            lsnf_panel_mma<WT>(gv[nt], gh1, lb, lane);
        });
#pragma unroll
        for (int t = 0; t < HT; ++t) gv[HT + t] = tp[t];
        if constexpr (DUMP) { if (live) lsnf_store_rows<HT>(gv, dmp + dl.off_gv, sample, a.nz, a.half, h, vec4); }
        // ---- B1: g_x = Wa gv ----
        lsnf_static_for<NZT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < NZT) lb = pipe.template acquire<NZT>(gb + C::OFF_B1 + (nt + 1) * NZT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<(SAVED ? 2 * HT : HT)>(gnext);
            gx[nt] = lsnf_zero16();
            lsnf_panel_mma<NZT>(gx[nt], gv, lb, lane);
        });
    }
    if (live && a.g_z_in) lsnf_store_rows<HT>(gx, a.g_z_in, sample, a.nz, a.half, h, vec4);
    if (a.z_new) {   // wave-uniform: fused Langevin update, tile by tile (keeps the register footprint flat)
        const float coef = 0.5f * a.step * a.step;
        float gf2 = 0.0f, gg2 = 0.0f;
        LsnfRngState rs = {0u, 0u, 0u, 0u, 0};
        if (!a.noise && a.rng.enabled) {
            const unsigned long long off = a.rng.offset + (a.rng.offset_dev ? *a.rng.offset_dev : 0ull);
            rs = {(unsigned)a.rng.seed, (unsigned)(a.rng.seed >> 32), (unsigned)off, (unsigned)(off >> 32), 1};
        }
        const float* zr = a.z_cur + row * (long)a.nz;
        float* zo = a.z_new + row * (long)a.nz;
#pragma unroll
        for (int t = 0; t < NZT; ++t) {
            const f32x16 zc = lsnf_load_tile<HT>(t, zr, a.half, h, vec4);
            f32x16 g = gx[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) gf2 += g[r] * g[r];
            if (a.grad_g) {
                const f32x16 gg = lsnf_load_tile<HT>(t, a.grad_g + row * (long)a.nz, a.half, h, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) { gg2 += gg[r] * gg[r]; g[r] = gg[r] + g[r]; }   // z_grad_g + z_grad_f (train.py:324)
            }
            f32x16 zn;
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[r] = zc[r] - coef * g[r];
            if (a.noise) {
                const f32x16 nv = lsnf_load_tile<HT>(t, a.noise + row * (long)a.nz, a.half, h, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = zn[r] + a.step * nv[r];                    // train.py:326
            } else if (rs.on) {
                const f32x16 nv = lsnf_noise_tile<HT>(t, (unsigned long long)(a.rng.row0 + sample), a.half, h, rs);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = zn[r] + a.step * nv[r];
            }
            if (live) lsnf_store_tile<HT>(t, zn, zo, a.half, h, vec4);
        }
        gf2 = lsnf_pair_sum(gf2);
        gg2 = lsnf_pair_sum(gg2);
        if (live && h == 0) {
            if (a.gf_norm) a.gf_norm[sample] = sqrtf(gf2);     // per-sample norms of train.py:328-329
            if (a.gg_norm) a.gg_norm[sample] = sqrtf(gg2);
        }
    }
}

template <class C, bool DUMP, int NW, bool SAVED>
hipError_t launch_bwd_w(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.depth * C::CONST_USED + 2 * (size_t)C::SLOT) * sizeof(float);
    auto kern = lsnf_bwd_z_kernel<C, DUMP, NW, SAVED>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * NW - 1) / (32 * NW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}
template <class C, bool DUMP>
hipError_t launch_bwd(const BwdArgs& a, hipStream_t stream) {
    // measured at B = 65536: 248 us with 4-wave workgroups, 263 us with 8 (more stages -> more, costlier 8-wave barriers)
    if constexpr (!DUMP) { if (a.act_saved) return launch_bwd_w<C, false, 4, true>(a, stream); }
    return launch_bwd_w<C, DUMP, 4, false>(a, stream);
}
}  // namespace

// dump == nullptr: plain backward w.r.t. z.  dump != nullptr: also writes the per-block intermediates
// (and accumulates sum dL/dlogdet into gl_total) that lsnf_params.hip turns into parameter gradients.
hipError_t lsnf_launch_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                  const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                  float* g_z_in, float* dump, float* gl_total, int vec4, hipStream_t stream,
                                  const LsnfLangevinArgs* lv, const float* act_saved) {
    BwdArgs a;
    a.act_saved = dump ? nullptr : act_saved;
    a.rng = LsnfRngArgs{0ull, 0ull, nullptr, 0ll, 0};
    a.dump = dump; a.gl_total = gl_total; a.width = g.width;
    a.z_cur = nullptr; a.grad_g = nullptr; a.noise = nullptr; a.z_new = nullptr; a.gf_norm = nullptr; a.gg_norm = nullptr; a.step = 0.f;
    if (lv) { a.z_cur = lv->z_cur; a.grad_g = lv->grad_g; a.noise = lv->noise; a.z_new = lv->z_new; a.gf_norm = lv->gf_norm;
              a.gg_norm = lv->gg_norm; a.step = lv->step; a.rng = lv->rng; }
    a.fwd_consts = plan + g.off_fwd_const; a.fwd_panels = plan + g.off_fwd_panels; a.bwd_panels = plan + g.off_bwd_panels;
    a.z_out = z_out; a.z_saved = z_saved; a.g_z1 = g_z1; a.g_logdet = g_logdet; a.g_z_in = g_z_in;
    a.ll_scale = ll_scale; a.ll_mode = ll_mode; a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (dump) {
        if (g.HT == 1 && g.WT == 1) return launch_bwd<BwdCfg<1, 1>, true>(a, stream);
        if (g.HT == 2 && g.WT == 2) return launch_bwd<BwdCfg<2, 2>, true>(a, stream);
        if (g.HT == 2 && g.WT == 4) return launch_bwd<BwdCfg<2, 4>, true>(a, stream);
    } else {
        if (g.HT == 1 && g.WT == 1) return launch_bwd<BwdCfg<1, 1>, false>(a, stream);
        if (g.HT == 2 && g.WT == 2) return launch_bwd<BwdCfg<2, 2>, false>(a, stream);
        if (g.HT == 2 && g.WT == 4) return launch_bwd<BwdCfg<2, 4>, false>(a, stream);
    }
    return hipErrorInvalidValue;
}
