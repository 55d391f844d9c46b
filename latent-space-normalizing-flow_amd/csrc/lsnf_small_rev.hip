// lsnf_small_rev.hip -- latency-oriented reverse (sampling) pass for small / medium batches.
// Same math and ABI entry point (lsnf_reverse dispatches on B) as lsnf_rev.hip, which replaces reference
// model.py:484-498 / :424-456; work decomposition of lsnf_small.h.
// Per block, last to first (x = [z1|z2] in LDS tiles X):
//   R2,R3,R4 : h = f(z1) -> [t; p]                                (forward panels S2..S4)
//   CI       : z2 = z2 / sigmoid(p) - t ; objective -= sum log sigmoid(p)     (model.py:436-438)
//   I1       : z = ([z1,z2] @ W^-1) * exp(-3 logs) - b ; objective -= log|det W| + sum 3 logs  (:193-196, 270, 246)
#include "lsnf_small.h"

namespace {

template <int HT_, int WT_>
struct SmallRevCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    using S::HT; using S::WT; using S::NZT;
    static constexpr int CONST_PER_BLOCK = S::FWD_CONST + S::INV_CONST;
    using R2 = SmallStage<WT, HT>;
    using R3 = SmallStage<WT, WT>;
    using R4 = SmallStage<2 * HT, WT>;
    using I1 = SmallStage<NZT, NZT, 1>;                        // unsplit: writes final tiles
    static constexpr int T_X = 0;                              // 2 x NZT ping-pong
    static constexpr int T_H1 = T_X + 2 * NZT;
    static constexpr int T_H2 = T_H1 + R2::OUT_TILES;
    static constexpr int T_TP = T_H2 + R3::OUT_TILES;
    static constexpr int T_U = T_TP + R4::OUT_TILES;           // NZT: [z1 | z2 after the inverse coupling] (final)
    static constexpr int T_END = T_U + NZT;
    static constexpr int AUX_FLOATS = 64 * (HT + 1);
};

struct SmallRevArgs {
    const float* fwd_consts; const float* fwd_panels; const float* inv_consts; const float* inv_panels;
    const float* z_in; const float* objective; float* z_out; float* objective_out;
    int B, nz, half, depth, vec4;
};

template <class C>
__global__ __launch_bounds__(LSNF_WG_THREADS, 1) void lsnf_small_rev_kernel(const SmallRevArgs a) {
    constexpr int HT = C::HT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tiles = smem;
    float* aux = smem + (size_t)C::T_END * LSNF_TILE_FLOATS;
    float* cst = aux + C::AUX_FLOATS;                                // depth * CONST_PER_BLOCK
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int vec4 = a.vec4;
    auto T = [&](int t) { return tiles + (size_t)t * LSNF_TILE_FLOATS; };

    const int last = a.depth - 1;
    // weights two stages ahead, issued between the running stage's MFMA groups (lsnf_small.h, SmallFetch)
    auto f2 = C::R2::fetch(a.fwd_panels + (size_t)last * C::FWD_BLOCK + C::OFF_S2, wave, lane);
    auto f3 = C::R3::fetch(a.fwd_panels + (size_t)last * C::FWD_BLOCK + C::OFF_S3, wave, lane);
    for (int i = tid; i < a.depth * C::CONST_PER_BLOCK; i += LSNF_WG_THREADS) {
        const int blk = i / C::CONST_PER_BLOCK, r = i % C::CONST_PER_BLOCK;
        cst[i] = r < C::FWD_CONST ? a.fwd_consts[blk * C::FWD_CONST + r] : a.inv_consts[blk * C::INV_CONST + (r - C::FWD_CONST)];
    }
    const long sample = (long)blockIdx.x * LSNF_SMALL_SAMPLES + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    if (wave < NZT) small_store_tile(T(C::T_X + (last & 1) * NZT + wave),
                                     lsnf_load_tile<HT>(wave, a.z_in + row * (long)a.nz, a.half, h, vec4), lane);
    float obj = 0.0f;                                                // carried by wave 0
    if (wave == 0) obj = a.objective ? a.objective[row] : 0.0f;
    __syncthreads();

    for (int blk = last; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_PER_BLOCK;
        const float* ci = cb + C::FWD_CONST;
        const float* gf = a.fwd_panels + (size_t)blk * C::FWD_BLOCK;
        const float* gi = a.inv_panels + (size_t)blk * C::INV_BLOCK;
        float* X = T(C::T_X + (blk & 1) * NZT);
        float* Xn = T(C::T_X + ((blk + 1) & 1) * NZT);
        float* H1 = T(C::T_H1); float* H2 = T(C::T_H2); float* TP = T(C::T_TP); float* U = T(C::T_U);
        auto tile = [&](const float* base, int t) { return small_load_tile(base + (size_t)t * LSNF_TILE_FLOATS, lane); };

        auto p4 = C::R4::begin_fetch(gf + C::OFF_S4, wave, lane);
        C::R2::run(f2, H1, wave, lane, [&](int kt) { return tile(X, kt); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + nt), h); }, p4);
        __syncthreads();
        auto pi = C::I1::begin_fetch(gi, wave, lane);
        C::R3::run(f3, H2, wave, lane,
                   [&](int kt) { return small_gather_tile<C::R2::KS, true>(H1 + (size_t)kt * C::R2::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + nt), h); }, pi);
        __syncthreads();
        const float* gfn = a.fwd_panels + (size_t)(blk > 0 ? blk - 1 : 0) * C::FWD_BLOCK;   // block 0 re-fetches its own panels
        auto p2 = C::R2::begin_fetch(gfn + C::OFF_S2, wave, lane);
        C::R4::run(p4.f, TP, wave, lane,
                   [&](int kt) { return small_gather_tile<C::R3::KS, true>(H2 + (size_t)kt * C::R3::KS * LSNF_TILE_FLOATS, lane); },
                   [&](int nt) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + nt), h); }, p2);
        f2 = p2.f;
        __syncthreads();
        // inverse coupling on waves 0..HT-1; waves HT..2HT-1 forward z1
        if (wave < HT) {
            const int j = wave;
            const f32x16 t = small_gather_tile<C::R4::KS, false>(TP + (size_t)j * C::R4::KS * LSNF_TILE_FLOATS, lane);
            const f32x16 p = small_gather_tile<C::R4::KS, false>(TP + (size_t)(HT + j) * C::R4::KS * LSNF_TILE_FLOATS, lane);
            f32x16 z2 = tile(X, HT + j);
            float lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(p[r], sig, lsig);
                z2[r] = z2[r] / sig - t[r];
                lsum += lsig;
            }
            small_store_tile(U + (size_t)(HT + j) * LSNF_TILE_FLOATS, z2, lane);
            aux[64 * j + lane] = lsum;
        } else if (wave < 2 * HT) {
            const int j = wave - HT;
            small_store_tile(U + (size_t)j * LSNF_TILE_FLOATS, tile(X, j), lane);
        }
        __syncthreads();
        if (wave == 0) {
            float ls = 0.0f;
#pragma unroll
            for (int j = 0; j < HT; ++j) ls += aux[64 * j + lane];
            obj = obj - lsnf_pair_sum(ls);
            obj = obj - cb[32 * C::NP + 1];   // logdet - log|det W|       (model.py:196)
            obj = obj - cb[32 * C::NP + 0];   // logdet - sum(3 logs)      (model.py:273-276, reverse)
        }
        auto p3 = C::R3::begin_fetch(gfn + C::OFF_S3, wave, lane);
        C::I1::run(pi.f, Xn, wave, lane, [&](int kt) { return tile(U, kt); }, [&](int nt) { return lsnf_bias_init(ci + 32 * nt, h); }, p3);
        f3 = p3.f;
        __syncthreads();
    }
    float* Xf = T(C::T_X + (1 & 1) * NZT);      // block 0 wrote X[(0+1)&1]
    if (wave < NZT && live)
        lsnf_store_tile<HT>(wave, small_load_tile(Xf + (size_t)wave * LSNF_TILE_FLOATS, lane), a.z_out + sample * (long)a.nz,
                            a.half, h, vec4);
    if (wave == 0 && live && h == 0 && a.objective_out) a.objective_out[sample] = obj;
}

template <class C>
hipError_t launch_small_rev(const SmallRevArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)C::T_END * LSNF_TILE_FLOATS + C::AUX_FLOATS + (size_t)a.depth * C::CONST_PER_BLOCK) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_small_rev_kernel<C>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + LSNF_SMALL_SAMPLES - 1) / LSNF_SMALL_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LSNF_WG_THREADS), lds, stream, a);
    return hipGetLastError();
}
}  // namespace

hipError_t lsnf_launch_small_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                     float* z_out, float* objective_out, int vec4, hipStream_t stream) {
    SmallRevArgs a;
    a.fwd_consts = plan + g.off_fwd_const; a.fwd_panels = plan + g.off_fwd_panels;
    a.inv_consts = plan + g.off_inv_const; a.inv_panels = plan + g.off_inv_panels;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.objective_out = objective_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small_rev<SmallRevCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small_rev<SmallRevCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small_rev<SmallRevCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
