// lsnf_small3_bwd.hip -- latency backward w.r.t. z (+ fused Langevin update) on the bf16 matrix pipe, from the forward's
// activation stash: the partner of lsnf_small3_fwd.hip (16-sample workgroups, L16 layout, producer-side bf16x3 split).
// Same math and ABI entry points as lsnf_small_bwd.hip's SAVED variant (replaces autograd of train.py:316-329).
//
// Per block, last to first; wave w owns the half-units g_x1[w], g_x2[w] of the running gradient (registers):
//   CB : s = sigma (stash); g_t = g_v2 = g_y2*s ; g_p = (1-s)(g_y2*y2 + g_l)            -> GTP, GV[second half]
//   B4 : g_a2 = (W3s g_t + W3p g_p) * [h2 > 0]   (masks: stash)                          -> GA2
//   B3 : g_a1 = (W2' g_a2) * [h1 > 0]                                                    -> GA1
//   B2 : g_v1 = g_x1 + W1' g_a1                                                          -> GV[first half]
//   B1 : g_x  = Wa [g_v1; g_v2]                                                          -> registers
// The transposed matrices come as bf16x3 panels in the 16x16x32 operand order (plan region off_b3b_panels).
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
    // LDS map (floats)
    static constexpr int L_GTP = 0;                                        // 2HT B-tiles: g_t | g_p
    static constexpr int L_GA2 = L_GTP + 2 * HT_ * S3_BTILE_FLOATS;
    static constexpr int L_GA1 = L_GA2 + WT_ * S3_BTILE_FLOATS;
    static constexpr int L_GV = L_GA1 + WT_ * S3_BTILE_FLOATS;             // NZT B-tiles: g_v1 | g_v2
    static constexpr int L_RED = L_GV + S::NZT * S3_BTILE_FLOATS;
    static constexpr int L_END = L_RED + 4 * 16 * 2;
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

template <class C, bool DUMP>
__global__ __launch_bounds__(256, 1) void lsnf_small3_bwd_kernel(const Small3BwdArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* GTP = smem + C::L_GTP;
    float* GA2 = smem + C::L_GA2;
    float* GA1 = smem + C::L_GA1;
    float* GV = smem + C::L_GV;
    float* RED = smem + C::L_RED;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int vec4 = a.vec4;

    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }

    const int last = a.depth - 1;
    const float* gb_last = a.panels + (size_t)last * C::BLOCKB;
    // weights two stages ahead: B4 and B3 of the last block first
    UFrags<2 * HT> wb4[NU2];
    UFrags<WT> wb3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) {
        wb4[i] = fetch_unit<2 * HT>(gb_last + C::OFFB4, hw[i] >> 1, hw[i] & 1, lane);
        wb3[i] = fetch_unit<WT>(gb_last + C::OFFB3, hw[i] >> 1, hw[i] & 1, lane);
    }

    const long sample = (long)blockIdx.x * S3_SAMPLES + n;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const size_t wtile = (size_t)(blockIdx.x >> 1);
    const int st = (int)(blockIdx.x & 1);
    const int lane32 = 16 * st + n + 32 * (g & 1);
    const int nibsh_base = 4 * (g >> 1);                                   // + 8*ft: bit offset of this lane's nibble in a mask word

    // this block's stash slice and block output (second half unit), one block ahead of their use
    f32x4 y2, sg; unsigned m1[NU2], m2[NU2];
    auto fetch_block_state = [&](int blk) {
        const float* act = a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile;
        const float* ysrc = (blk == a.depth - 1) ? a.z_out + row * (long)a.nz : a.z_saved + ((size_t)blk * a.B + row) * a.nz;
        y2 = load_row_half<HT>(HT + nt1, ft1, ysrc, a.half, g, vec4);
        sg = reinterpret_cast<const f32x4*>(act + (size_t)nt1 * 1024)[(2 * ft1 + (g >> 1)) * 64 + lane32];
        const unsigned* words = reinterpret_cast<const unsigned*>(act + al.mask_off);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            m1[i] = (words[nt * 64 + lane32] >> (nibsh_base + 8 * ft)) & 0xFu;
            m2[i] = (words[(WT + nt) * 64 + lane32] >> (nibsh_base + 8 * ft)) & 0xFu;
        }
    };
    fetch_block_state(last);

    // upstream gradient on the stack's output, this wave's two half-units
    float gl;
    f32x4 gx1, gx2;
    {
        const float* zo = a.z_out + row * (long)a.nz;
        if (a.ll_mode) {            // L = ll_scale * sum ll: dL/dz1 = -ll_scale * z1, dL/dlogdet = ll_scale (train.py:317-320)
            gl = a.ll_scale;
            const f32x4 y1 = load_row_half<HT>(nt1, ft1, zo, a.half, g, vec4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { gx1[r] = -a.ll_scale * y1[r]; gx2[r] = -a.ll_scale * y2[r]; }
        } else {
            gl = a.g_logdet ? a.g_logdet[row] : 0.0f;
            if (a.g_z1) {
                gx1 = load_row_half<HT>(nt1, ft1, a.g_z1 + row * (long)a.nz, a.half, g, vec4);
                gx2 = load_row_half<HT>(HT + nt1, ft1, a.g_z1 + row * (long)a.nz, a.half, g, vec4);
            } else { gx1 = zero4(); gx2 = zero4(); }
        }
    }

    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const bool w4 = (a.width & 3) == 0, h4 = (a.half & 3) == 0;
    if constexpr (DUMP) {   // G = sum_b dL/dlogdet_b: one atomic per workgroup
        if (wave == 0) {
            float t = (live && g == 0) ? gl : 0.0f;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) atomicAdd(a.gl_total, t);
        }
    }

    for (int blk = last; blk >= 0; --blk) {
        const float* gb = a.panels + (size_t)blk * C::BLOCKB;
        const int nb = blk > 0 ? blk - 1 : 0;                              // block 0 re-fetches its own panels: no loads under a branch
        const float* gbn = a.panels + (size_t)nb * C::BLOCKB;
        float* dmp = (DUMP && live) ? a.dump + (size_t)blk * dl.per_block : nullptr;

        // ---- CB: coupling backward (model.py:414-418) on this wave's half-unit ----
        if (has1) {
            f32x4 gt, gp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gt[r] = gx2[r] * sg[r];
                gp[r] = (1.0f - sg[r]) * (gx2[r] * y2[r] + gl);
            }
            store_half(GTP + nt1 * S3_BTILE_FLOATS, ft1, gt, lane);
            store_half(GTP + (HT + nt1) * S3_BTILE_FLOATS, ft1, gp, lane);
            store_half(GV + (HT + nt1) * S3_BTILE_FLOATS, ft1, gt, lane);       // g_v2 = g_t
            if constexpr (DUMP) {
                if (dmp) {
                    store_plain_half(gt, dmp + dl.off_gt + sample * (long)a.half, a.half, nt1, ft1, g, h4);
                    store_plain_half(gp, dmp + dl.off_gp + sample * (long)a.half, a.half, nt1, ft1, g, h4);
                    store_row_half<HT>(HT + nt1, ft1, gt, dmp + dl.off_gv + sample * (long)a.nz, a.half, g, vec4);
                }
            }
        }
        __syncthreads();
        // ---- B4: g_a2 = ([W3s W3p][g_t; g_p]) gated by h2 > 0 ----
        UFrags<WT> wb2 = fetch_unit<WT>(gb + C::OFFB2, nt1, ft1, lane);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const f32x4 ga = mask4(unit_mma<2 * HT>(zero4(), wb4[i], GTP, lane), m2[i]);
            if (hasw[i]) {
                store_half(GA2 + (hw[i] >> 1) * S3_BTILE_FLOATS, hw[i] & 1, ga, lane);
                if constexpr (DUMP) { if (dmp) store_plain_half(ga, dmp + dl.off_ga2 + sample * (long)a.width, a.width, hw[i] >> 1, hw[i] & 1, g, w4); }
            }
        }
        __syncthreads();
        // ---- B3: g_a1 = (W2' g_a2) gated by h1 > 0 ----
        UFrags<NZT> wb1a = fetch_unit<NZT>(gb + C::OFFB1, nt1, ft1, lane);
        UFrags<NZT> wb1b = fetch_unit<NZT>(gb + C::OFFB1, HT + nt1, ft1, lane);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const f32x4 ga = mask4(unit_mma<WT>(zero4(), wb3[i], GA2, lane), m1[i]);
            if (hasw[i]) {
                store_half(GA1 + (hw[i] >> 1) * S3_BTILE_FLOATS, hw[i] & 1, ga, lane);
                if constexpr (DUMP) { if (dmp) store_plain_half(ga, dmp + dl.off_ga1 + sample * (long)a.width, a.width, hw[i] >> 1, hw[i] & 1, g, w4); }
            }
        }
        __syncthreads();
        // ---- B2: g_v1 = g_x1 (direct) + W1' g_a1 ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) wb4[i] = fetch_unit<2 * HT>(gbn + C::OFFB4, hw[i] >> 1, hw[i] & 1, lane);
        fetch_block_state(nb);                                             // next block's y2 / sigma / masks (this block's are consumed)
        {
            const f32x4 gv1 = unit_mma<WT>(gx1, wb2, GA1, lane);
            if (has1) {
                store_half(GV + nt1 * S3_BTILE_FLOATS, ft1, gv1, lane);
                if constexpr (DUMP) { if (dmp) store_row_half<HT>(nt1, ft1, gv1, dmp + dl.off_gv + sample * (long)a.nz, a.half, g, vec4); }
            }
        }
        __syncthreads();
        // ---- B1: g_x = Wa [g_v1; g_v2] ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) wb3[i] = fetch_unit<WT>(gbn + C::OFFB3, hw[i] >> 1, hw[i] & 1, lane);
        gx1 = unit_mma<NZT>(zero4(), wb1a, GV, lane);
        gx2 = unit_mma<NZT>(zero4(), wb1b, GV, lane);
        __syncthreads();
    }

    // ---- outputs: g_z_in and / or the fused Langevin update (train.py:324-329) ----
    float gf2 = 0.0f, gg2 = 0.0f;
    if (has1) {
        if (live && a.g_z_in) {
            float* gr = a.g_z_in + sample * (long)a.nz;
            store_row_half<HT>(nt1, ft1, gx1, gr, a.half, g, vec4);
            store_row_half<HT>(HT + nt1, ft1, gx2, gr, a.half, g, vec4);
        }
        if (a.z_new) {
            const float coef = 0.5f * a.step * a.step;
            LsnfRngState rs = {0u, 0u, 0u, 0u, 0};
            if (!a.noise && a.rng.enabled) {
                const unsigned long long off = a.rng.offset + (a.rng.offset_dev ? *a.rng.offset_dev : 0ull);
                rs = {(unsigned)a.rng.seed, (unsigned)(a.rng.seed >> 32), (unsigned)off, (unsigned)(off >> 32), 1};
            }
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {                               // this wave's first-half and second-half unit
                const int t = hs * HT + nt1;
                const f32x4 gq = hs ? gx2 : gx1;
                const f32x4 zc = load_row_half<HT>(t, ft1, a.z_cur + row * (long)a.nz, a.half, g, vec4);
                f32x4 gs = gq;
#pragma unroll
                for (int r = 0; r < 4; ++r) gf2 += gq[r] * gq[r];
                if (a.grad_g) {
                    const f32x4 gg = load_row_half<HT>(t, ft1, a.grad_g + row * (long)a.nz, a.half, g, vec4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gg2 += gg[r] * gg[r]; gs[r] = gg[r] + gq[r]; }   // z_grad_g + z_grad_f (train.py:324)
                }
                f32x4 zn;
#pragma unroll
                for (int r = 0; r < 4; ++r) zn[r] = zc[r] - coef * gs[r];
                if (a.noise) {
                    const f32x4 nv = load_row_half<HT>(t, ft1, a.noise + row * (long)a.nz, a.half, g, vec4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) zn[r] = zn[r] + a.step * nv[r];                    // train.py:326
                } else if (rs.on) {     // the same draws as every other kernel: a function of (seed, offset, global row, column)
                    const unsigned long long grow = (unsigned long long)(a.rng.row0 + sample);
                    const int f0 = 32 * nt1 + 16 * ft1 + 4 * g;
                    unsigned c0 = ((unsigned)hs << 16) | (unsigned)(f0 >> 2), c1 = (unsigned)grow, c2 = rs.c2, c3 = rs.c3hi ^ (unsigned)(grow >> 32);
                    lsnf_philox4x32_10(c0, c1, c2, c3, rs.k0, rs.k1);
                    float n0, n1, n2, n3;
                    lsnf_box_muller(c0, c1, n0, n1);
                    lsnf_box_muller(c2, c3, n2, n3);
                    zn[0] += a.step * n0; zn[1] += a.step * n1; zn[2] += a.step * n2; zn[3] += a.step * n3;
                }
                if (live) store_row_half<HT>(t, ft1, zn, a.z_new + sample * (long)a.nz, a.half, g, vec4);
            }
        }
    }
    if (a.z_new && (a.gf_norm || a.gg_norm)) {      // kernel-uniform: per-sample norms of train.py:328-329
        gf2 = group_sum(gf2); gg2 = group_sum(gg2);
        if (g == 0) { RED[(wave * 16 + n) * 2] = gf2; RED[(wave * 16 + n) * 2 + 1] = gg2; }
        __syncthreads();
        if (wave == 0 && g == 0 && live) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += RED[(w * 16 + n) * 2]; s2 += RED[(w * 16 + n) * 2 + 1]; }
            if (a.gf_norm) a.gf_norm[sample] = sqrtf(s1);
            if (a.gg_norm) a.gg_norm[sample] = sqrtf(s2);
        }
    }
}

template <class C>
hipError_t launch_small3_bwd(const Small3BwdArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)C::L_END * sizeof(float);
    auto kern = a.dump ? lsnf_small3_bwd_kernel<C, true> : lsnf_small3_bwd_kernel<C, false>;
    static unsigned long long lds_ok[2] = {0, 0};
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok[a.dump ? 1 : 0]); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + S3_SAMPLES - 1) / S3_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
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
