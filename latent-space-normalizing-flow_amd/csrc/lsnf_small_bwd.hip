// lsnf_small_bwd.hip -- latency-oriented backward w.r.t. z (+ fused Langevin update) for small / medium batches.
// Same math and ABI entry points (lsnf_backward_z, lsnf_langevin_step dispatch on B) as lsnf_bwd.hip, which replaces
// autograd of train.py:316-329; work decomposition of lsnf_small.h (32 samples per workgroup, stages split over
// the 4 waves, tiles exchanged through LDS, weights straight from L2 one stage ahead).
//
// Per block, last to first (g = dL/d[z1|z2'] lives in LDS tiles GX, block output y = [v1|y2] in Y):
//   R2,R3,R4 : recompute h1, h2, [t;p] from v1                       (forward panels S2..S4)
//   CB       : s = sigmoid(p); g_t = g_v2 = g_y2*s ; g_p = (1-s)(g_y2*y2 + g_l)
//   B4       : g_h2 = [W3s W3p][g_t;g_p]        B3 : g_h1 = W2'(g_h2 * [h2>0])
//   B2       : g_v1 = g_v1 + W1'(g_h1 * [h1>0]) B1 : g_x  = Wa [g_v1; g_v2]
// SAVED variant (the forward filled act_saved): R2..R4 disappear, sigma and the relu masks come from the stash,
// and CB is folded into B4's operand fetch -- 4 stages per block instead of 8.
// DUMP variant (parameter gradients, lsnf_params.hip): the recomputing kernel additionally writes, per block, the
// per-sample tensors of LsnfDumpLayout (h1, h2, g_a1, g_a2, g_t, g_p, g_v) -- each tile by the one wave that
// owns output tile 0 of the stage consuming it -- and accumulates sum_b dL/dlogdet_b.
#include "lsnf_small.h"

namespace {

template <int HT_, int WT_>
struct SmallBwdCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    using S::HT; using S::WT; using S::NZT;
    static constexpr int CONST_USED = 32 * (S::P2 + S::P3 + S::P4);     // S2, S3, S4 biases
    using R2 = SmallStage<WT, HT>;
    using R3 = SmallStage<WT, WT>;
    using R4 = SmallStage<2 * HT, WT>;
    using B4 = SmallStage<WT, 2 * HT>;
    using B3 = SmallStage<WT, WT>;
    using B2 = SmallStage<HT, WT, 1>;                          // unsplit: writes final g_v1 tiles
    using B1 = SmallStage<NZT, NZT, 1>;                        // unsplit: writes final g_x tiles
    // LDS map (tiles)
    static constexpr int T_GX = 0;                             // 2 x NZT ping-pong running gradient
    static constexpr int T_Y = T_GX + 2 * NZT;                 // NZT block output
    static constexpr int T_H1 = T_Y + NZT;
    static constexpr int T_H2 = T_H1 + R2::OUT_TILES;
    static constexpr int T_TP = T_H2 + R3::OUT_TILES;          // [t;p] partials; dead after CB -> reused for g_h2 partials
    static constexpr int T_GH2 = T_TP;
    static constexpr int N_TP = R4::OUT_TILES > B4::OUT_TILES ? R4::OUT_TILES : B4::OUT_TILES;
    static constexpr int T_GTP = T_TP + N_TP;                  // 2HT: g_t tiles then g_p tiles (final); dead after B4 -> g_h1 partials
    static constexpr int T_GH1 = T_GTP;
    static constexpr int N_GTP = 2 * HT > B3::OUT_TILES ? 2 * HT : B3::OUT_TILES;
    static constexpr int T_GV = T_GTP + N_GTP;                 // NZT: [g_v1 ; g_v2] (final)
    static constexpr int T_END = T_GV + NZT;
    static constexpr int AUX_FLOATS = 64 * (2 * NZT + 2);
};

struct SmallBwdArgs {
    const float* fwd_consts; const float* fwd_panels; const float* bwd_panels;
    const float* z_out; const float* z_saved; const float* act_saved; const float* g_z1; const float* g_logdet;
    float* g_z_in;
    float* dump; float* gl_total;      // DUMP variant only
    const float* z_cur; const float* grad_g; const float* noise; float* z_new; float* gf_norm; float* gg_norm;
    float step, ll_scale;
    LsnfRngArgs rng;                   // Langevin update: in-kernel noise when `noise` is NULL and rng.enabled
    int ll_mode, B, nz, half, width, depth, vec4;
};

// one plain-pad tile (feature f = 32*t + o(r,h), valid f < ncols) -> dense (B, ncols) row-major
__device__ __forceinline__ void small_store_plain(const f32x16& x, float* __restrict__ dst, long row, int ncols, int t, int h) {
    float* d = dst + row * (long)ncols;
    const bool v4 = (ncols & 3) == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * t + 8 * g + 4 * h;
        if (v4) {
            if (f0 < ncols) { f32x4 v = {x[4 * g + 0], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]}; *reinterpret_cast<f32x4*>(d + f0) = v; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < ncols) d[f0 + j] = x[4 * g + j];
        }
    }
}

template <class C, bool SAVED, bool DUMP>
__global__ __launch_bounds__(LSNF_WG_THREADS, 1) void lsnf_small_bwd_kernel(const SmallBwdArgs a) {
    static_assert(!(DUMP && SAVED), "the parameter-gradient dump needs the recomputed activations");
    constexpr int HT = C::HT, NZT = C::NZT, WT = C::WT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tiles = smem;
    float* aux = smem + (size_t)C::T_END * LSNF_TILE_FLOATS;
    float* cst = aux + C::AUX_FLOATS;                                // depth * CONST_USED
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int vec4 = a.vec4;
    auto T = [&](int t) { return tiles + (size_t)t * LSNF_TILE_FLOATS; };

    const int last = a.depth - 1;
    // weights travel L2/HBM -> VGPR two stages ahead of their use, issued between the MFMA groups of the running
    // stage (lsnf_small.h, SmallFetch).  In flight at loop entry: R2, R3 of the last block, or (SAVED) its B4, B3;
    // the pair the variant does not use is dead code.
    auto f2 = C::R2::fetch(a.fwd_panels + (size_t)last * C::FWD_BLOCK + C::OFF_S2, wave, lane);
    auto f3 = C::R3::fetch(a.fwd_panels + (size_t)last * C::FWD_BLOCK + C::OFF_S3, wave, lane);
    auto fb4 = C::B4::fetch(a.bwd_panels + (size_t)last * C::BWD_BLOCK + C::OFF_B4, wave, lane);
    auto fb3 = C::B3::fetch(a.bwd_panels + (size_t)last * C::BWD_BLOCK + C::OFF_B3, wave, lane);
    // SAVED: this wave's slice of the stash, one block ahead: sigma tiles, h2 masks of its B3 slice, h1 masks
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    constexpr int K3 = C::B3::KTL;
    const int ks3 = (wave < C::B3::UNITS) ? wave % C::B3::KS : 0;
    f32x16 sg[HT];
    unsigned mk1[WT], mk2[K3];
    auto fetch_act = [&](int blk) {
        const float* act = a.act_saved + (size_t)blk * al.per_block + (size_t)blockIdx.x * al.per_tile;
#pragma unroll
        for (int j = 0; j < HT; ++j) sg[j] = lsnf_act_load_sigma(act, j, lane);
#pragma unroll
        for (int k = 0; k < WT; ++k) mk1[k] = *lsnf_act_mask_ptr(act, al.mask_off, k, lane);
#pragma unroll
        for (int k = 0; k < K3; ++k) mk2[k] = *lsnf_act_mask_ptr(act, al.mask_off, WT + ks3 * K3 + k, lane);
    };
    if constexpr (SAVED) fetch_act(last);
    for (int i = tid; i < a.depth * C::CONST_USED; i += LSNF_WG_THREADS) {
        const int blk = i / C::CONST_USED, r = i % C::CONST_USED;
        cst[i] = a.fwd_consts[blk * C::FWD_CONST + 32 * C::P1 + r];
    }
    const long sample = (long)blockIdx.x * LSNF_SMALL_SAMPLES + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;

    // upstream gradient -> GX[parity of last block], block output of the last block -> Y
    float gl;
    if (a.ll_mode) gl = a.ll_scale; else gl = a.g_logdet ? a.g_logdet[row] : 0.0f;
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    if constexpr (DUMP) {   // G = sum_b dL/dlogdet_b : one atomic per workgroup
        if (wave == 0) {
            float t = (live && h == 0) ? gl : 0.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) atomicAdd(a.gl_total, t);
        }
    }
    if (wave < NZT) {
        const f32x16 y = lsnf_load_tile<HT>(wave, a.z_out + row * (long)a.nz, a.half, h, vec4);
        small_store_tile(T(C::T_Y + wave), y, lane);
        f32x16 g;
        if (a.ll_mode) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = -a.ll_scale * y[r];     // dL/dz1 = -ll_scale * z1 (train.py:317-320)
        } else if (a.g_z1) g = lsnf_load_tile<HT>(wave, a.g_z1 + row * (long)a.nz, a.half, h, vec4);
        else g = lsnf_zero16();
        small_store_tile(T(C::T_GX + (last & 1) * NZT + wave), g, lane);
    }
    __syncthreads();

    for (int blk = last; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_USED;
        const float* gf = a.fwd_panels + (size_t)blk * C::FWD_BLOCK;
        const float* gb = a.bwd_panels + (size_t)blk * C::BWD_BLOCK;
        float* GX = T(C::T_GX + (blk & 1) * NZT);
        float* GXn = T(C::T_GX + ((blk + 1) & 1) * NZT);           // becomes GX of block blk-1
        float* Y = T(C::T_Y); float* H1 = T(C::T_H1); float* H2 = T(C::T_H2); float* TP = T(C::T_TP);
        float* GTP = T(C::T_GTP); float* GH2 = T(C::T_GH2); float* GH1 = T(C::T_GH1); float* GV = T(C::T_GV);
        auto tile = [&](const float* base, int t) { return small_load_tile(base + (size_t)t * LSNF_TILE_FLOATS, lane); };
        float* dmp = DUMP ? a.dump + (size_t)blk * dl.per_block : nullptr;

        if constexpr (!SAVED) {
        // ---- R2: h1 = W1'^T v1 + c1 ----
        auto p4 = C::R4::begin_fetch(gf + C::OFF_S4, wave, lane);
        C::R2::run(f2, H1, wave, lane, [&](int kt) { return tile(Y, kt); }, [&](int nt) { return lsnf_bias_init(cb + 32 * nt, h); }, p4);
        __syncthreads();
        // ---- R3: h2 = W2'^T relu(h1) + c2 ----
        auto pb4 = C::B4::begin_fetch(gb + C::OFF_B4, wave, lane);
        C::R3::run(f3, H2, wave, lane,
                   [&](int kt) { return small_gather_tile<C::R2::KS, true>(H1 + (size_t)kt * C::R2::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P2 + nt), h); }, pb4);
        __syncthreads();
        // ---- R4: [t; p] ----
        auto pb3 = C::B3::begin_fetch(gb + C::OFF_B3, wave, lane);
        C::R4::run(p4.f, TP, wave, lane,
                   [&](int kt) { return small_gather_tile<C::R3::KS, true>(H2 + (size_t)kt * C::R3::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P2 + C::P3 + nt), h); }, pb3);
        fb4 = pb4.f; fb3 = pb3.f;
        __syncthreads();
        // ---- CB: coupling backward on waves 0..HT-1 ----
        if (wave < HT) {
            const int j = wave;
            const f32x16 p = small_gather_tile<C::R4::KS, false>(TP + (size_t)(HT + j) * C::R4::KS * LSNF_TILE_FLOATS, lane);
            const f32x16 gy2 = tile(GX, HT + j), y2 = tile(Y, HT + j);
            f32x16 gt, gp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(p[r], sig, lsig);
                gt[r] = gy2[r] * sig;
                gp[r] = (1.0f - sig) * (gy2[r] * y2[r] + gl);
            }
            small_store_tile(GTP + (size_t)j * LSNF_TILE_FLOATS, gt, lane);
            small_store_tile(GTP + (size_t)(HT + j) * LSNF_TILE_FLOATS, gp, lane);
            small_store_tile(GV + (size_t)(HT + j) * LSNF_TILE_FLOATS, gt, lane);
            if constexpr (DUMP) {
                if (live) { small_store_plain(gt, dmp + dl.off_gt, sample, a.half, j, h); small_store_plain(gp, dmp + dl.off_gp, sample, a.half, j, h); }
            }
        }
        __syncthreads();
        }   // !SAVED
        // ---- B4: g_h2 = [W3s W3p][g_t; g_p] ----
        auto pb2 = C::B2::begin_fetch(gb + C::OFF_B2, wave, lane);
        if constexpr (SAVED) {
            // operand tile kt (< HT: g_t_kt, else g_p_{kt-HT}) built on the fly from the stashed sigma; the wave that
            // owns output tile 0 also leaves g_v2 = g_t in GV for B1.  kt % HT == k % HT for every split of B4.
            const bool owner = (wave / C::B4::KS) == 0;
            C::B4::run(fb4, GH2, wave, lane,
                       [&](int kt, int k) {
                           const int j = kt < HT ? kt : kt - HT;                 // wave-uniform
                           const f32x16 gy2 = tile(GX, HT + j);
                           const f32x16& s = sg[k % HT];
                           f32x16 o;
                           if (kt < HT) {
#pragma unroll
                               for (int r = 0; r < 16; ++r) o[r] = gy2[r] * s[r];
                               if (owner) small_store_tile(GV + (size_t)(HT + j) * LSNF_TILE_FLOATS, o, lane);
                           } else {
                               const f32x16 y2 = tile(Y, HT + j);
#pragma unroll
                               for (int r = 0; r < 16; ++r) o[r] = (1.0f - s[r]) * (gy2[r] * y2[r] + gl);
                           }
                           return o;
                       },
                       [&](int) { return lsnf_zero16(); }, pb2);
        } else {
            C::B4::run(fb4, GH2, wave, lane, [&](int kt) { return tile(GTP, kt); }, [&](int) { return lsnf_zero16(); }, pb2);
        }
        __syncthreads();
        // ---- B3: g_h1 = W2' (g_h2 gated by h2 > 0) ----
        auto pb1 = C::B1::begin_fetch(gb + C::OFF_B1, wave, lane);
        C::B3::run(fb3, GH1, wave, lane,
                   [&](int kt, int k) {
                       const f32x16 gh = small_gather_tile<C::B4::KS, false>(GH2 + (size_t)kt * C::B4::KS * LSNF_TILE_FLOATS, lane);
                       if constexpr (SAVED) return lsnf_apply_mask16(gh, mk2[k]);
                       else {
                           const f32x16 pre = small_gather_tile<C::R3::KS, false>(H2 + (size_t)kt * C::R3::KS * LSNF_TILE_FLOATS, lane);
                           const f32x16 ga = small_gate16(gh, pre);
                           if constexpr (DUMP) {
                               if (live && wave / C::B3::KS == 0) {
                                   small_store_plain(lsnf_relu16(pre), dmp + dl.off_h2, sample, a.width, kt, h);
                                   small_store_plain(ga, dmp + dl.off_ga2, sample, a.width, kt, h);
                               }
                           }
                           return ga;
                       }
                   },
                   [&](int) { return lsnf_zero16(); }, pb1);
        __syncthreads();
        // ---- B2: g_v1 = g_v1(direct) + W1' (g_h1 gated by h1 > 0) -> GV[0..HT) ----
        const int nb = blk > 0 ? blk - 1 : 0;      // block 0 re-fetches its own panels: no loads under a branch
        // next block's output tile: global load issued here, parked in registers until Y is free
        // (unconditional for the same reason: idle waves / block 0 load a valid tile and drop it)
        const float* ysrc = blk > 0 ? a.z_saved + ((size_t)(blk - 1) * a.B + row) * a.nz : a.z_out + row * (long)a.nz;
        const f32x16 ynext = lsnf_load_tile<HT>(wave < NZT ? wave : 0, ysrc, a.half, h, vec4);
        auto run_b2 = [&](auto& pn) {
        C::B2::run(pb2.f, GV, wave, lane,
                   [&](int kt, int k) {
                       const f32x16 gh = small_gather_tile<C::B3::KS, false>(GH1 + (size_t)kt * C::B3::KS * LSNF_TILE_FLOATS, lane);
                       if constexpr (SAVED) return lsnf_apply_mask16(gh, mk1[k]);
                       else {
                           const f32x16 pre = small_gather_tile<C::R2::KS, false>(H1 + (size_t)kt * C::R2::KS * LSNF_TILE_FLOATS, lane);
                           const f32x16 ga = small_gate16(gh, pre);
                           if constexpr (DUMP) {
                               if (live && wave == 0) {       // B2 is unsplit: wave 0 owns output tile 0
                                   small_store_plain(lsnf_relu16(pre), dmp + dl.off_h1, sample, a.width, kt, h);
                                   small_store_plain(ga, dmp + dl.off_ga1, sample, a.width, kt, h);
                               }
                           }
                           return ga;
                       }
                   },
                   [&](int nt) { return tile(GX, nt); }, pn);
        };
        if constexpr (SAVED) { auto pn = C::B4::begin_fetch(a.bwd_panels + (size_t)nb * C::BWD_BLOCK + C::OFF_B4, wave, lane); run_b2(pn); fb4 = pn.f; }
        else { auto pn = C::R2::begin_fetch(a.fwd_panels + (size_t)nb * C::FWD_BLOCK + C::OFF_S2, wave, lane); run_b2(pn); f2 = pn.f; }
        __syncthreads();
        // ---- B1: g_x = Wa [g_v1; g_v2] -> GXn ----
        if constexpr (SAVED) fetch_act(nb);
        auto run_b1 = [&](auto& pn) {
        C::B1::run(pb1.f, GXn, wave, lane,
                   [&](int kt) {
                       const f32x16 gvt = tile(GV, kt);
                       if constexpr (DUMP) {
                           if (live && wave == 0) lsnf_store_tile<HT>(kt, gvt, dmp + dl.off_gv + sample * (long)a.nz, a.half, h, vec4);
                       }
                       return gvt;
                   },
                   [&](int) { return lsnf_zero16(); }, pn);
        };
        if constexpr (SAVED) { auto pn = C::B3::begin_fetch(a.bwd_panels + (size_t)nb * C::BWD_BLOCK + C::OFF_B3, wave, lane); run_b1(pn); fb3 = pn.f; }
        else { auto pn = C::R3::begin_fetch(a.fwd_panels + (size_t)nb * C::FWD_BLOCK + C::OFF_S3, wave, lane); run_b1(pn); f3 = pn.f; }
        if (blk > 0 && wave < NZT) small_store_tile(Y + (size_t)wave * LSNF_TILE_FLOATS, ynext, lane);   // Y's readers are done
        __syncthreads();
    }

    // ---- outputs: g_z_in and / or the fused Langevin update (train.py:324-329), one tile per wave ----
    float* GXf = T(C::T_GX + (1 & 1) * NZT);   // block 0 wrote GXn = GX[(0+1)&1]
    float gf2 = 0.0f, gg2 = 0.0f;
    if (wave < NZT) {
        const f32x16 g = small_load_tile(GXf + (size_t)wave * LSNF_TILE_FLOATS, lane);
        if (live && a.g_z_in) lsnf_store_tile<HT>(wave, g, a.g_z_in + sample * (long)a.nz, a.half, h, vec4);
        if (a.z_new) {
            const float coef = 0.5f * a.step * a.step;
            const f32x16 zc = lsnf_load_tile<HT>(wave, a.z_cur + row * (long)a.nz, a.half, h, vec4);
            f32x16 gs = g;
#pragma unroll
            for (int r = 0; r < 16; ++r) gf2 += g[r] * g[r];
            if (a.grad_g) {
                const f32x16 gg = lsnf_load_tile<HT>(wave, a.grad_g + row * (long)a.nz, a.half, h, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) { gg2 += gg[r] * gg[r]; gs[r] = gg[r] + g[r]; }
            }
            f32x16 zn;
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[r] = zc[r] - coef * gs[r];
            if (a.noise) {
                const f32x16 nv = lsnf_load_tile<HT>(wave, a.noise + row * (long)a.nz, a.half, h, vec4);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = zn[r] + a.step * nv[r];
            } else if (a.rng.enabled) {
                const unsigned long long off = a.rng.offset + (a.rng.offset_dev ? *a.rng.offset_dev : 0ull);
                const LsnfRngState rs = {(unsigned)a.rng.seed, (unsigned)(a.rng.seed >> 32), (unsigned)off, (unsigned)(off >> 32), 1};
                const f32x16 nv = lsnf_noise_tile<HT>(wave, (unsigned long long)(a.rng.row0 + sample), a.half, h, rs);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = zn[r] + a.step * nv[r];
            }
            if (live) lsnf_store_tile<HT>(wave, zn, a.z_new + sample * (long)a.nz, a.half, h, vec4);
        }
    }
    if (a.z_new && (a.gf_norm || a.gg_norm)) {      // kernel-uniform
        if (wave < NZT) { aux[64 * wave + lane] = gf2; aux[64 * (NZT + wave) + lane] = gg2; }
        __syncthreads();
        if (wave == 0) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int t = 0; t < NZT; ++t) { s1 += aux[64 * t + lane]; s2 += aux[64 * (NZT + t) + lane]; }
            s1 = lsnf_pair_sum(s1); s2 = lsnf_pair_sum(s2);
            if (live && h == 0) {
                if (a.gf_norm) a.gf_norm[sample] = sqrtf(s1);
                if (a.gg_norm) a.gg_norm[sample] = sqrtf(s2);
            }
        }
    }
}

template <class C, bool SAVED, bool DUMP>
hipError_t launch_small_bwd_v(const SmallBwdArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)C::T_END * LSNF_TILE_FLOATS + C::AUX_FLOATS + (size_t)a.depth * C::CONST_USED) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_small_bwd_kernel<C, SAVED, DUMP>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + LSNF_SMALL_SAMPLES - 1) / LSNF_SMALL_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LSNF_WG_THREADS), lds, stream, a);
    return hipGetLastError();
}
template <class C>
hipError_t launch_small_bwd(const SmallBwdArgs& a, hipStream_t stream) {
    if (a.dump) return launch_small_bwd_v<C, false, true>(a, stream);
    return a.act_saved ? launch_small_bwd_v<C, true, false>(a, stream) : launch_small_bwd_v<C, false, false>(a, stream);
}
}  // namespace

// returns hipErrorInvalidValue when the geometry's LDS footprint does not fit (caller falls back to lsnf_bwd.hip)
hipError_t lsnf_launch_small_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                        const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale, float* g_z_in,
                                        int vec4, hipStream_t stream, const LsnfLangevinArgs* lv, const float* act_saved,
                                        float* dump, float* gl_total) {
    SmallBwdArgs a;
    a.act_saved = dump ? nullptr : act_saved;
    a.dump = dump; a.gl_total = gl_total; a.width = g.width;
    a.fwd_consts = plan + g.off_fwd_const; a.fwd_panels = plan + g.off_fwd_panels; a.bwd_panels = plan + g.off_bwd_panels;
    a.z_out = z_out; a.z_saved = z_saved; a.g_z1 = g_z1; a.g_logdet = g_logdet; a.g_z_in = g_z_in;
    a.rng = LsnfRngArgs{0ull, 0ull, nullptr, 0ll, 0};
    a.z_cur = nullptr; a.grad_g = nullptr; a.noise = nullptr; a.z_new = nullptr; a.gf_norm = nullptr; a.gg_norm = nullptr; a.step = 0.f;
    if (lv) { a.z_cur = lv->z_cur; a.grad_g = lv->grad_g; a.noise = lv->noise; a.z_new = lv->z_new; a.gf_norm = lv->gf_norm;
              a.gg_norm = lv->gg_norm; a.step = lv->step; a.rng = lv->rng; }
    a.ll_scale = ll_scale; a.ll_mode = ll_mode; a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small_bwd<SmallBwdCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small_bwd<SmallBwdCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small_bwd<SmallBwdCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
