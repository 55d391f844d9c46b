// lsnf_bwd3.hip -- throughput backward w.r.t. z (+ fused Langevin update) from the forward's activation stash, on the bf16
// matrix pipe: the bf16x3 split and L16 lane layout of lsnf_fwd3.hip's 16x16x32 kernel applied to lsnf_bwd.hip's SAVED
// variant (replaces autograd of train.py:316-329).  Per block, last to first, a wave owning 32 samples and all features:
//   CB : s = sigma (stash); g_t = g_v2 = g_y2*s ; g_p = (1-s)(g_y2*y2 + g_l)
//   B4 : g_a2 = ([W3s W3p][g_t; g_p]) * [h2 > 0]       B3 : g_a1 = (W2' g_a2) * [h1 > 0]        (masks: stash)
//   B2 : g_v1 = g_x1 + W1' g_a1                          B1 : g_x  = Wa [g_v1; g_v2]
// Transposed matrices as bf16x3 panels in the 16x16x32 operand order (plan region off_b3b_panels), streamed L2 -> LDS in
// panel pairs as in the forward.
#include <stdlib.h>
#include "lsnf_l16.h"

namespace {

template <int HT_, int WT_>
struct Bwd3Cfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = LSNF_FRAG3_FLOATS;
    static constexpr int OFFB4 = 0;
    static constexpr int OFFB3 = OFFB4 + F * WT_ * 2 * HT_;
    static constexpr int OFFB2 = OFFB3 + F * WT_ * WT_;
    static constexpr int OFFB1 = OFFB2 + F * HT_ * WT_;
    static constexpr int BLOCKB = OFFB1 + F * S::NZT * S::NZT;
    static constexpr int MAXKT = (S::NZT > WT_ ? S::NZT : WT_);
    static constexpr int SLOT3 = 2 * MAXKT * F;
};

struct Bwd3Args {
    const float* panels;
    const float* z_out; const float* z_saved; const float* act_saved; const float* g_z1; const float* g_logdet;
    float* g_z_in;
    const float* z_cur; const float* grad_g; const float* noise; float* z_new; float* gf_norm; float* gg_norm;
    float step, ll_scale;
    LsnfRngArgs rng;
    int ll_mode, B, nz, half, depth, vec4;
    float* dump; float* gl_total; int width; int dump_tiled;     // dump_tiled: the g arrays in the tiled form (lsnf_l16.h l16_store_tiled), g_v as its first half only
         // DUMP variant (parameter gradients, lsnf_params.hip): per block g_v, g_a1, g_a2,
                                                  // g_t, g_p written for the batch contraction; G = sum_b dL/dlogdet_b
};

// stash -> L16 registers (inverse of l16_store_sigma / l16_store_masks)
__device__ __forceinline__ f32x16 l16_load_sigma(const float* tile_base, int t, int n, int g) {
    const f32x4* p = reinterpret_cast<const f32x4*>(tile_base + (size_t)t * 1024);
    f32x16 a;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const f32x4 v = p[(2 * ft + (g >> 1)) * 64 + 16 * st + n + 32 * (g & 1)];
            const int b = (2 * ft + st) * 4;
            a[b] = v[0]; a[b + 1] = v[1]; a[b + 2] = v[2]; a[b + 3] = v[3];
        }
    return a;
}
__device__ __forceinline__ unsigned l16_load_mask(const unsigned* words, int n, int g) {   // bit (2*ft+st)*4 + r
    unsigned m = 0;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const unsigned w = words[16 * st + n + 32 * (g & 1)];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) m |= ((w >> (8 * ft + 4 * (g >> 1))) & 0xFu) << ((2 * ft + st) * 4);
    }
    return m;
}

template <class C, int NW, int DUMP>       // DUMP: 0 none, 1 parameter-gradient dump row-major, 2 tiled (lsnf_l16.h l16_store_tiled)
__global__ __launch_bounds__(64 * NW, 1) void lsnf_bwd3_kernel(const Bwd3Args a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;                                        // 2 x SLOT3
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int vec4 = a.vec4;

    const int last = a.depth - 1;
    Pipe3<NW> pipe;
    pipe.buf0 = buf0; pipe.slot = C::SLOT3; pipe.wave = wave; pipe.lane = lane;
    pipe.template prime<first_kib(WT, 2 * HT)>(a.panels + (size_t)last * C::BLOCKB + C::OFFB4);

    const long base = ((long)blockIdx.x * NW + wave) * 32;
    long sample[2], rows[2]; bool live[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) { sample[st] = base + 16 * st + n; live[st] = sample[st] < a.B; rows[st] = live[st] ? sample[st] : (long)a.B - 1; }
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    size_t wtile = (size_t)blockIdx.x * NW + wave;                   // clamped: waves past the batch read a valid tile, store nothing
    if (wtile * 32 >= (size_t)a.B) wtile = (size_t)(a.B - 1) / 32;

    // upstream gradient on the stack's output
    f32x16 gx[NZT];
    float gl[2];
    if (a.ll_mode) {
#pragma unroll
        for (int t = 0; t < NZT; ++t) {
            const f32x16 y = l16_load_tile<HT>(t, a.z_out, rows, a.nz, a.half, g, vec4);
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[t][r] = -a.ll_scale * y[r];        // dL/dz1 = -ll_scale * z1 (train.py:317-320)
        }
        gl[0] = gl[1] = a.ll_scale;
    } else {
#pragma unroll
        for (int t = 0; t < NZT; ++t) gx[t] = a.g_z1 ? l16_load_tile<HT>(t, a.g_z1, rows, a.nz, a.half, g, vec4) : lsnf_zero16();
#pragma unroll
        for (int st = 0; st < 2; ++st) gl[st] = a.g_logdet ? a.g_logdet[rows[st]] : 0.0f;
    }
    auto zero = [](int) { return lsnf_zero16(); };
    auto keep = [](f32x16 acc, int) { return acc; };
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    // tiled dump: this wave's 32-sample tile (not the clamped one: a wave past the batch stores nothing).  Rows past the batch inside
    // the last tile must read as zeros (they are inside the arrays and the contraction sums whole tiles): the upstream gradient of a
    // dead row is zeroed below, and every dumped quantity is linear in it.
    const size_t tile32 = (size_t)blockIdx.x * NW + wave;
    const bool tile_ok = tile32 * 32 < (size_t)a.B;
    if constexpr (DUMP == 2) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (!live[st]) gl[st] = 0.0f;
#pragma unroll
            for (int t = 0; t < NZT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (((r >> 2) & 1) == st && !live[st]) gx[t][r] = 0.0f;
        }
    }
    const bool w4 = (a.width & 3) == 0, h4 = (a.half & 3) == 0;
    if constexpr (DUMP) {   // G = sum_b dL/dlogdet_b: one atomic per wave
        float t = 0.0f;
#pragma unroll
        for (int st = 0; st < 2; ++st) t += (live[st] && g == 0) ? gl[st] : 0.0f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) atomicAdd(a.gl_total, t);
    }

    for (int blk = last; blk >= 0; --blk) {
        const float* gb = a.panels + (size_t)blk * C::BLOCKB;
        const float* gnext = blk > 0 ? a.panels + (size_t)(blk - 1) * C::BLOCKB + C::OFFB4 : nullptr;
        const float* ysrc = (blk == last) ? a.z_out : a.z_saved + (size_t)blk * a.B * a.nz;
        const float* act = a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile;
        const unsigned* words = reinterpret_cast<const unsigned*>(act + al.mask_off);
        unsigned m1[WT], m2[WT];
#pragma unroll
        for (int t = 0; t < WT; ++t) { m1[t] = l16_load_mask(words + t * 64, n, g); m2[t] = l16_load_mask(words + (WT + t) * 64, n, g); }

        // ---- CB: coupling backward (model.py:414-418) with the stashed sigma: tp[0..HT) <- g_t (= g_v2), tp[HT..2HT) <- g_p ----
        f32x16 tp[2 * HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            const f32x16 y2 = l16_load_tile<HT>(HT + t, ysrc, rows, a.nz, a.half, g, vec4);
            const f32x16 sg = l16_load_sigma(act, t, n, g);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gy2 = gx[HT + t][r];
                tp[t][r] = gy2 * sg[r];
                tp[HT + t][r] = (1.0f - sg[r]) * (gy2 * y2[r] + gl[(r >> 2) & 1]);
            }
        }
        float* dmp = DUMP ? a.dump + (size_t)blk * dl.per_block : nullptr;
        if constexpr (DUMP) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                if constexpr (DUMP == 2) {
                    if (tile_ok) { l16_store_tiled(tp[t], dmp + dl.off_gt, tile32, a.half, t, n, g); l16_store_tiled(tp[HT + t], dmp + dl.off_gp, tile32, a.half, t, n, g); }
                } else {
                    l16_store_plain(tp[t], dmp + dl.off_gt, sample, live, a.half, t, g, h4);
                    l16_store_plain(tp[HT + t], dmp + dl.off_gp, sample, live, a.half, t, g, h4);
                }
            }
        }
        // ---- B4: g_a2 = ([W3s W3p][g_t; g_p]) gated by h2 > 0 ----
        f32x16 gh2[WT];
        {
            Split3 ts[4 * HT];
            l16_split_tiles<2 * HT>(tp, ts);
            l16_gemm_stage3<WT, 2 * HT, first_kib(WT, WT)>(pipe, gb + C::OFFB4, gb + C::OFFB3, gh2, ts, zero, keep);
        }
#pragma unroll
        for (int t = 0; t < WT; ++t) gh2[t] = lsnf_apply_mask16(gh2[t], m2[t]);
        if constexpr (DUMP) {
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if constexpr (DUMP == 2) { if (tile_ok) l16_store_tiled(gh2[t], dmp + dl.off_ga2, tile32, a.width, t, n, g); }
                else l16_store_plain(gh2[t], dmp + dl.off_ga2, sample, live, a.width, t, g, w4);
            }
        }
        // ---- B3: g_a1 = (W2' g_a2) gated by h1 > 0 ----
        f32x16 gh1[WT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(gh2, hs);
            l16_gemm_stage3<WT, WT, first_kib(HT, WT)>(pipe, gb + C::OFFB3, gb + C::OFFB2, gh1, hs, zero, keep);
        }
#pragma unroll
        for (int t = 0; t < WT; ++t) gh1[t] = lsnf_apply_mask16(gh1[t], m1[t]);
        if constexpr (DUMP) {
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if constexpr (DUMP == 2) { if (tile_ok) l16_store_tiled(gh1[t], dmp + dl.off_ga1, tile32, a.width, t, n, g); }
                else l16_store_plain(gh1[t], dmp + dl.off_ga1, sample, live, a.width, t, g, w4);
            }
        }
        // ---- B2: g_v1 = g_x1 (direct) + W1' g_a1 ;  gv = [g_v1 ; g_v2] ----
        f32x16 gv[NZT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(gh1, hs);
            l16_gemm_stage3<HT, WT, first_kib(NZT, NZT)>(pipe, gb + C::OFFB2, gb + C::OFFB1, gv, hs,
                                                         [&](int t) { return gx[t]; }, keep);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) gv[HT + t] = tp[t];
        if constexpr (DUMP) {
#pragma unroll
            for (int t = 0; t < NZT; ++t) {
                // (tiled: g_v1 only, as a (B, half) array -- the second half IS g_t, which the contraction reads from its own array)
                if constexpr (DUMP == 2) { if (t < HT && tile_ok) l16_store_tiled(gv[t], dmp + dl.off_gv, tile32, a.half, t, n, g); }
                else l16_store_tile<HT>(t, gv[t], dmp + dl.off_gv, sample, live, a.nz, a.half, g, vec4);
            }
        }
        // ---- B1: g_x = Wa [g_v1; g_v2] ----
        {
            Split3 vs[2 * NZT];
            l16_split_tiles<NZT>(gv, vs);
            l16_gemm_stage3<NZT, NZT, first_kib(WT, 2 * HT)>(pipe, gb + C::OFFB1, gnext, gx, vs, zero, keep);
        }
    }

    // ---- outputs: g_z_in and / or the fused Langevin update (train.py:324-329), tile by tile ----
    if (a.g_z_in) {
#pragma unroll
        for (int t = 0; t < NZT; ++t) l16_store_tile<HT>(t, gx[t], a.g_z_in, sample, live, a.nz, a.half, g, vec4);
    }
    if (a.z_new) {   // kernel-uniform
        const float coef = 0.5f * a.step * a.step;
        float gf2[2] = {0.f, 0.f}, gg2[2] = {0.f, 0.f};
        LsnfRngState rs = {0u, 0u, 0u, 0u, 0};
        if (!a.noise && a.rng.enabled) {
            const unsigned long long off = a.rng.offset + (a.rng.offset_dev ? *a.rng.offset_dev : 0ull);
            rs = {(unsigned)a.rng.seed, (unsigned)(a.rng.seed >> 32), (unsigned)off, (unsigned)(off >> 32), 1};
        }
#pragma unroll
        for (int t = 0; t < NZT; ++t) {
            const f32x16 zc = l16_load_tile<HT>(t, a.z_cur, rows, a.nz, a.half, g, vec4);
            f32x16 gs = gx[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) gf2[(r >> 2) & 1] += gx[t][r] * gx[t][r];
            if (a.grad_g) {
                const f32x16 gg = l16_load_tile<HT>(t, a.grad_g, rows, a.nz, a.half, g, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) { gg2[(r >> 2) & 1] += gg[r] * gg[r]; gs[r] = gg[r] + gx[t][r]; }   // train.py:324
            }
            f32x16 zn;
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[r] = zc[r] - coef * gs[r];
            if (a.noise) {
                const f32x16 nv = l16_load_tile<HT>(t, a.noise, rows, a.nz, a.half, g, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = zn[r] + a.step * nv[r];                                    // train.py:326
            } else if (rs.on) {     // the same draws as every other kernel: a function of (seed, offset, global row, column)
                const int hh = t / HT, tt = t % HT;
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const unsigned long long grow = (unsigned long long)(a.rng.row0 + sample[st]);
                        const int f0 = 32 * tt + 16 * ft + 4 * g, b = (2 * ft + st) * 4;
                        unsigned c0 = ((unsigned)hh << 16) | (unsigned)(f0 >> 2), c1 = (unsigned)grow, c2 = rs.c2, c3 = rs.c3hi ^ (unsigned)(grow >> 32);
                        lsnf_philox4x32_10(c0, c1, c2, c3, rs.k0, rs.k1);
                        float n0, n1, n2, n3;
                        lsnf_box_muller(c0, c1, n0, n1);
                        lsnf_box_muller(c2, c3, n2, n3);
                        zn[b] += a.step * n0; zn[b + 1] += a.step * n1; zn[b + 2] += a.step * n2; zn[b + 3] += a.step * n3;
                    }
            }
            l16_store_tile<HT>(t, zn, a.z_new, sample, live, a.nz, a.half, g, vec4);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float s1 = l16_group_sum(gf2[st]), s2 = l16_group_sum(gg2[st]);
            if (live[st] && g == 0) {
                if (a.gf_norm) a.gf_norm[sample[st]] = sqrtf(s1);     // per-sample norms of train.py:328-329
                if (a.gg_norm) a.gg_norm[sample[st]] = sqrtf(s2);
            }
        }
    }
}

template <class C, int NW>
hipError_t launch_bwd3_w(const Bwd3Args& a, hipStream_t stream) {
    const size_t lds = 2 * (size_t)C::SLOT3 * sizeof(float);
    auto kern = a.dump ? (a.dump_tiled ? lsnf_bwd3_kernel<C, NW, 2> : lsnf_bwd3_kernel<C, NW, 1>) : lsnf_bwd3_kernel<C, NW, 0>;
    static unsigned long long lds_ok[3] = {0, 0, 0};
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok[a.dump ? (a.dump_tiled ? 2 : 1) : 0]); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * NW - 1) / (32 * NW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}
template <class C>
hipError_t launch_bwd3(const Bwd3Args& a, hipStream_t stream) {
    static const char* fw = getenv("LSNF_FORCE_WAVES");   // experiment knob (tools/): 4 or 8
    const bool eight = fw ? atoi(fw) == 8 : a.B > 128 * 256;
    return eight ? launch_bwd3_w<C, 8>(a, stream) : launch_bwd3_w<C, 4>(a, stream);
}
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); needs the activation stash; hipErrorInvalidValue = not covered.
// The f_width-128 instantiation lives in a second translation unit (lsnf_bwd3w.hip = this file with LSNF_BWD3_WIDE_TU) that is
// compiled WITHOUT packed fp32 math: measured at B = 65 536 (tools/ab_secondary.py, three builds alternating in one job) the
// (2,4) kernel is 2 % faster that way (186 vs 190 us), the (2,2) kernel 16 % slower (125.5 vs 108.1 us) -- v_pk_add_f32 /
// v_pk_mul_f32 wait for the MFMA pipe (tools/micro/shadow.hip) but halve the instruction count of the split, and which
// effect wins depends on the kernel's phase structure.
#ifdef LSNF_BWD3_WIDE_TU
#define LSNF_BWD3_ENTRY lsnf_launch_backward3_z_wide
#else
#define LSNF_BWD3_ENTRY lsnf_launch_backward3_z
hipError_t lsnf_launch_backward3_z_wide(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                        const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                        float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                        float* dump, float* gl_total, int dump_tiled);
#endif
hipError_t LSNF_BWD3_ENTRY(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                           const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                           float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                           float* dump, float* gl_total, int dump_tiled) {
    if (!act_saved) return hipErrorInvalidValue;
    Bwd3Args a;
    a.dump = dump; a.gl_total = gl_total; a.width = g.width; a.dump_tiled = dump_tiled;
    a.panels = plan + g.off_b3b_panels;
    a.z_out = z_out; a.z_saved = z_saved; a.act_saved = act_saved; a.g_z1 = g_z1; a.g_logdet = g_logdet; a.g_z_in = g_z_in;
    a.z_cur = nullptr; a.grad_g = nullptr; a.noise = nullptr; a.z_new = nullptr; a.gf_norm = nullptr; a.gg_norm = nullptr; a.step = 0.f;
    a.rng = LsnfRngArgs{0ull, 0ull, nullptr, 0ll, 0};
    if (lv) { a.z_cur = lv->z_cur; a.grad_g = lv->grad_g; a.noise = lv->noise; a.z_new = lv->z_new; a.gf_norm = lv->gf_norm;
              a.gg_norm = lv->gg_norm; a.step = lv->step; a.rng = lv->rng; }
    a.ll_scale = ll_scale; a.ll_mode = ll_mode; a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
#ifdef LSNF_BWD3_WIDE_TU
    if (g.HT == 2 && g.WT == 4) return launch_bwd3<Bwd3Cfg<2, 4>>(a, stream);
#else
    if (g.HT == 1 && g.WT == 1) return launch_bwd3<Bwd3Cfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_bwd3<Bwd3Cfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4)
        return lsnf_launch_backward3_z_wide(g, plan, B, z_out, z_saved, act_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4, stream, lv, dump, gl_total, dump_tiled);
#endif
    return hipErrorInvalidValue;
}
