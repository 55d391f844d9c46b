// lsnf_small_fwd.hip -- latency-oriented forward of the flow stack for small / medium batches.
// Same math, same prepared weights and same ABI call (lsnf_forward dispatches on B) as lsnf_fwd.hip, which
// replaces reference model.py:473-483 + train.py:317-319; see lsnf_small.h for the work decomposition:
// one workgroup = 32 samples, its 4 waves split every GEMM stage, activations exchanged through LDS.
#include "lsnf_small.h"

namespace {

template <int HT_, int WT_>
struct SmallFwdCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    using S::HT; using S::WT; using S::NZT;
    static constexpr int BLOCK_FLOATS = S::FWD_BLOCK, CONST_FLOATS = S::FWD_CONST;
    using S1 = SmallStage<NZT, NZT>;                           // v  = Wa^T x + ca
    using S2 = SmallStage<WT, HT>;                             // h1 = W1'^T v1 + c1        (relu applied by the reader)
    using S3 = SmallStage<WT, WT>;                             // h2 = W2'^T relu(h1) + c2
    using S4 = SmallStage<2 * HT, WT>;                         // [t; p] = W3^T relu(h2) + c3
    // LDS map (tiles of LSNF_TILE_FLOATS)
    static constexpr int T_X = 0;                              // 2 x NZT  (ping-pong block input / output)
    static constexpr int T_V = T_X + 2 * NZT;                  // S1 out
    static constexpr int T_H1 = T_V + S1::OUT_TILES;
    static constexpr int T_H2 = T_H1 + S2::OUT_TILES;
    static constexpr int T_TP = T_H2 + S3::OUT_TILES;
    static constexpr int T_END = T_TP + S4::OUT_TILES;
    static constexpr int AUX_FLOATS = 64 * (HT + NZT + 2);     // log-scale partials, sum-of-squares partials, reduction scratch
};

struct SmallFwdArgs {
    const float* consts; const float* panels;
    const float* z_in; const float* objective;
    float* z_out; float* logdet_out; float* ll_out; float* z_saved;
    float* act_saved;      // NULL or the activation stash (already offset to first_block), lsnf_layout.h LsnfActLayout
    double* stats;
    int B, nz, half, n_blocks, vec4;
    unsigned long long* stamps;        // LSNF_STAMPS diagnostic build only: [grid][4 waves][64] shader-clock stamps
};

#ifdef LSNF_STAMPS
#define SMALL_STAMP(i)                                                                                  \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * 4 + wave) & 2047) * 64 + (i)] = t_; } while (0)
#else
#define SMALL_STAMP(i) do {} while (0)
#endif

template <class C>
__global__ __launch_bounds__(LSNF_WG_THREADS, 1) void lsnf_small_fwd_kernel(const SmallFwdArgs a) {
    constexpr int HT = C::HT, NZT = C::NZT, WT = C::WT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tiles = smem;                                            // C::T_END tiles
    float* aux = smem + (size_t)C::T_END * LSNF_TILE_FLOATS;         // AUX_FLOATS
    float* cst = aux + C::AUX_FLOATS;                                // n_blocks * CONST_FLOATS
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int vec4 = a.vec4;

    SMALL_STAMP(0);
    // weights travel L2/HBM -> VGPR TWO stages ahead of their use (a fetch costs ~3000 cycles when the line comes
    // from HBM -- the L2 starts every launch cold -- and a stage lasts 1000..4000)
    auto f1 = C::S1::fetch(a.panels, wave, lane);
    auto f2 = C::S2::fetch(a.panels + C::OFF_S2, wave, lane);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += LSNF_WG_THREADS) cst[i] = a.consts[i];

    const long sample = (long)blockIdx.x * LSNF_SMALL_SAMPLES + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    if (wave < NZT) small_store_tile(tiles + (size_t)(C::T_X + wave) * LSNF_TILE_FLOATS,
                                     lsnf_load_tile<HT>(wave, a.z_in + row * (long)a.nz, a.half, h, vec4), lane);
    float ell = 0.0f;                                               // carried by wave 0
    if (wave == 0) ell = a.objective ? a.objective[row] : 0.0f;
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    __syncthreads();
    SMALL_STAMP(1);

    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels + (size_t)blk * C::BLOCK_FLOATS;
        const bool more = blk + 1 < a.n_blocks;
        float* X = tiles + (size_t)(C::T_X + (blk & 1) * NZT) * LSNF_TILE_FLOATS;
        float* Xn = tiles + (size_t)(C::T_X + ((blk + 1) & 1) * NZT) * LSNF_TILE_FLOATS;
        float* V = tiles + (size_t)C::T_V * LSNF_TILE_FLOATS;
        float* H1 = tiles + (size_t)C::T_H1 * LSNF_TILE_FLOATS;
        float* H2 = tiles + (size_t)C::T_H2 * LSNF_TILE_FLOATS;
        float* TP = tiles + (size_t)C::T_TP * LSNF_TILE_FLOATS;
        float* act = a.act_saved ? a.act_saved + (size_t)blk * al.per_block + (size_t)blockIdx.x * al.per_tile : nullptr;

        // ---- S1 (actnorm + 1x1 conv, model.py:244,268,187); weights of S2 fetched meanwhile ----
        auto p3 = C::S3::begin_fetch(gblk + C::OFF_S3, wave, lane);
        C::S1::run(f1, V, wave, lane, [&](int kt) { return small_load_tile(X + (size_t)kt * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * nt, h); }, p3);
        if (wave == 0) {   // logdet += sum(3 logs) ; += log|det W|   (model.py:273-276, 182, 189)
            ell = ell + cb[32 * C::NP + 0];
            ell = ell + cb[32 * C::NP + 1];
        }
        SMALL_STAMP(2 + 10 * blk + 0);
        __syncthreads();
        SMALL_STAMP(2 + 10 * blk + 1);
        // ---- S2 (model.py:326-328) ----
        auto p4 = C::S4::begin_fetch(gblk + C::OFF_S4, wave, lane);
        C::S2::run(f2, H1, wave, lane,
                   [&](int kt) { return small_gather_tile<C::S1::KS, false>(V + (size_t)kt * C::S1::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + nt), h); }, p4);
        SMALL_STAMP(2 + 10 * blk + 2);
        __syncthreads();
        SMALL_STAMP(2 + 10 * blk + 3);
        // ---- S3 ----
        // unconditional (the last block re-fetches its own panels): a fetch under `if (more)` would make the
        // s_waitcnt of this stage conservative, see SmallStage::fetch
        const float* gnext = more ? gblk + C::BLOCK_FLOATS : gblk;
        auto p1 = C::S1::begin_fetch(gnext, wave, lane);
        if (act && wave < WT)   // relu mask of h1 tile `wave` for the backward (model.py:307)
            *lsnf_act_mask_ptr(act, al.mask_off, wave, lane) =
                lsnf_posmask16(small_gather_tile<C::S2::KS, false>(H1 + (size_t)wave * C::S2::KS * LSNF_TILE_FLOATS, lane));
        C::S3::run(p3.f, H2, wave, lane,
                   [&](int kt) { return small_gather_tile<C::S2::KS, true>(H1 + (size_t)kt * C::S2::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + nt), h); }, p1);
        f1 = p1.f;
        SMALL_STAMP(2 + 10 * blk + 4);
        __syncthreads();
        SMALL_STAMP(2 + 10 * blk + 5);
        // ---- S4 (fc_zeros, shift / pre-sigmoid de-interleaved, model.py:347-349,411-413) ----
        auto p2 = C::S2::begin_fetch(gnext + C::OFF_S2, wave, lane);
        if (act && wave < WT)
            *lsnf_act_mask_ptr(act, al.mask_off, WT + wave, lane) =
                lsnf_posmask16(small_gather_tile<C::S3::KS, false>(H2 + (size_t)wave * C::S3::KS * LSNF_TILE_FLOATS, lane));
        C::S4::run(p4.f, TP, wave, lane,
                   [&](int kt) { return small_gather_tile<C::S3::KS, true>(H2 + (size_t)kt * C::S3::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + nt), h); }, p2);
        f2 = p2.f;
        SMALL_STAMP(2 + 10 * blk + 6);
        __syncthreads();
        SMALL_STAMP(2 + 10 * blk + 7);
        // ---- coupling (model.py:414-418): waves 0..HT-1 produce y2 tiles + log-scale partials,
        //      waves HT..2HT-1 forward the v1 tiles (concat, model.py:422) ----
        if (wave < HT) {
            const int j = wave;
            const f32x16 v2 = small_gather_tile<C::S1::KS, false>(V + (size_t)(HT + j) * C::S1::KS * LSNF_TILE_FLOATS, lane);
            const f32x16 t = small_gather_tile<C::S4::KS, false>(TP + (size_t)j * C::S4::KS * LSNF_TILE_FLOATS, lane);
            const f32x16 p = small_gather_tile<C::S4::KS, false>(TP + (size_t)(HT + j) * C::S4::KS * LSNF_TILE_FLOATS, lane);
            f32x16 y, sg;
            float lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, l2;
                lsnf_sigmoid_log2(p[r], sig, l2);
                y[r] = (v2[r] + t[r]) * sig;
                sg[r] = sig;
                lsum += l2;
            }
            if (act) lsnf_act_store_sigma(act, j, sg, lane);
            small_store_tile(Xn + (size_t)(HT + j) * LSNF_TILE_FLOATS, y, lane);
            aux[64 * j + lane] = lsum;
        } else if (wave < 2 * HT) {
            const int j = wave - HT;
            small_store_tile(Xn + (size_t)j * LSNF_TILE_FLOATS,
                             small_gather_tile<C::S1::KS, false>(V + (size_t)j * C::S1::KS * LSNF_TILE_FLOATS, lane), lane);
        }
        SMALL_STAMP(2 + 10 * blk + 8);
        __syncthreads();
        SMALL_STAMP(2 + 10 * blk + 9);
        if (wave == 0) {              // logdet += sum_j log(scale_j)
            float ls = 0.0f;
#pragma unroll
            for (int j = 0; j < HT; ++j) ls += aux[64 * j + lane];
            ell = ell + -0.6931471805599453f * lsnf_pair_sum(ls);
        }
        if (a.z_saved != nullptr && more && wave < NZT && live)
            lsnf_store_tile<HT>(wave, small_load_tile(Xn + (size_t)wave * LSNF_TILE_FLOATS, lane),
                                a.z_saved + ((size_t)blk * a.B + sample) * a.nz, a.half, h, vec4);
    }

    // ---- epilogue: z_out, logdet, ll (train.py:317-319) ----
    float* Xf = tiles + (size_t)(C::T_X + (a.n_blocks & 1) * NZT) * LSNF_TILE_FLOATS;
    float* ssb = aux + 64 * HT;
    if (wave < NZT) {
        const f32x16 x = small_load_tile(Xf + (size_t)wave * LSNF_TILE_FLOATS, lane);
        float ss = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += x[r] * x[r];
        ssb[64 * wave + lane] = ss;
        if (live) lsnf_store_tile<HT>(wave, x, a.z_out + sample * (long)a.nz, a.half, h, vec4);
    }
    __syncthreads();
    if (wave == 0) {
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < NZT; ++t) ss += ssb[64 * t + lane];
        ss = lsnf_pair_sum(ss);
        const float ll = (-0.5f * ss + 1.8378770664093453f) + ell;
        if (live && h == 0) {
            a.logdet_out[sample] = ell;
            if (a.ll_out) a.ll_out[sample] = ll;
        }
        if (a.stats) {
            double dl = (live && h == 0) ? (double)ll : 0.0, dd = (live && h == 0) ? (double)ell : 0.0;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
            lsnf_publish_stats(a.stats, dl, dd, a.B, lane);       // (wave-level protocol: all 64 lanes of wave 0)
        }
    }
    SMALL_STAMP(60);
}

template <class C>
hipError_t launch_small_fwd(const SmallFwdArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)C::T_END * LSNF_TILE_FLOATS + C::AUX_FLOATS + (size_t)a.n_blocks * C::CONST_FLOATS) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_small_fwd_kernel<C>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + LSNF_SMALL_SAMPLES - 1) / LSNF_SMALL_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LSNF_WG_THREADS), lds, stream, a);
    return hipGetLastError();
}
}  // namespace

hipError_t lsnf_launch_small_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                     const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                     float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                     hipStream_t stream) {
    SmallFwdArgs a;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels = plan + g.off_fwd_panels + (size_t)first_block * g.fwd_block_floats;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.act_saved = act_saved ? act_saved + (size_t)first_block * lsnf_act_layout(B, g.HT, g.WT).per_block : nullptr;
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    { extern unsigned long long* g_lsnf_stamps;
      if (!g_lsnf_stamps) { if (hipMalloc(&g_lsnf_stamps, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) g_lsnf_stamps = nullptr; }
      a.stamps = g_lsnf_stamps; }
#endif
    a.z_saved = z_saved; a.stats = stats; a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small_fwd<SmallFwdCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small_fwd<SmallFwdCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small_fwd<SmallFwdCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
