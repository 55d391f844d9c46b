// lsnf_fwd.hip -- fused forward of the flow-prior stack: ONE launch runs n_blocks coupling blocks
// (actnorm -> invertible 1x1 -> split -> 3-layer scale/shift MLP -> affine coupling -> log-det)
// plus the log-prob epilogue, with the latent rows resident in VGPRs from load to store.
//
// Replaces, per row of z:  reference model.py:473-483 (_netF.forward) -> :357-360 (revnet2d)
// -> :391-422 (revnet2d_step forward) -> :235-294 actnorm, :179-191 1x1 conv, :306-350 MLP,
// :411-418 coupling + log-scale reduction, and train.py:317-319 (log-prob assembly).
//
// Work decomposition: workgroup = 4 waves (one per SIMD), wave = 32 samples (sample on lane&31).
// Every GEMM runs on v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (output feature
// on the row index) and the ACTIVATIONS as the B operand, so a GEMM's accumulator registers are
// directly the next GEMM's B operand (lsnf_layout.h).  Weights stream L2 -> LDS by LDS-DMA in
// panels (one n-tile x all k-tiles, <= 16 KiB), double-buffered, one barrier per panel.
#include "lsnf_device.h"

namespace {

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

template <int HT_, int WT_>
struct FwdCfg {
    static constexpr int HT = HT_, WT = WT_, NZT = 2 * HT_;
    static constexpr int P1 = NZT, P2 = WT, P3 = WT, P4 = 2 * HT;  // panels per stage
    static constexpr int NP = P1 + P2 + P3 + P4;
    static constexpr int KT1 = NZT, KT2 = HT, KT3 = WT, KT4 = WT;
    static constexpr int MAXKT = (NZT > WT ? NZT : WT);
    static constexpr int SLOT = MAXKT * LSNF_FRAG_FLOATS;
    static constexpr int BLOCK_FLOATS = LSNF_FRAG_FLOATS * (P1 * KT1 + P2 * KT2 + P3 * KT3 + P4 * KT4);
    static constexpr int CONST_FLOATS = 32 * NP + 32;
    static constexpr int kt_of(int p) { return p < P1 ? KT1 : (p < P1 + P2 ? KT2 : KT3); }
    static constexpr int off_of(int p) {  // float offset of panel p inside a block's stream
        int o = 0;
        for (int q = 0; q < p; ++q) o += kt_of(q) * LSNF_FRAG_FLOATS;
        return o;
    }
};

struct FwdArgs {
    const float* consts;   // already offset to first_block
    const float* panels;   // already offset to first_block
    const float* z_in;
    const float* objective;
    float* z_out;
    float* logdet_out;
    float* ll_out;
    float* z_saved;
    int B, nz, half, n_blocks, vec4;
};

template <class C>
__global__ __launch_bounds__(LSNF_WG_THREADS, 2) void lsnf_fwd_kernel(const FwdArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                         // n_blocks * CONST_FLOATS
    float* buf0 = smem + a.n_blocks * C::CONST_FLOATS;         // 2 x SLOT
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 31, h = lane >> 5;

    // prologue: first panel in flight, constants to LDS, latent rows to registers
    lsnf_issue_panel<C::kt_of(0)>(a.panels, buf0, wave, lane);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += LSNF_WG_THREADS) cst[i] = a.consts[i];

    const long sample = ((long)blockIdx.x * LSNF_WG_WAVES + wave) * 32 + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;

    f32x16 x[NZT];
#ifdef LSNF_ABLATE_IO    // timing diagnostic only: no HBM reads of z (prices the exposed prologue load)
#pragma unroll
    for (int t = 0; t < NZT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = 0.001f * (float)(lane + r + t);
    float ell = 0.0f;
#else
    lsnf_load_rows<HT>(x, a.z_in, row, a.nz, a.half, h, a.vec4 != 0);
    float ell = a.objective ? a.objective[row] : 0.0f;
#endif

    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels + (size_t)blk * C::BLOCK_FLOATS;
        const bool more = blk + 1 < a.n_blocks;

        // acquire<P>: panel P of this block is ready in buffer P&1 after this; prefetch P+1.
        auto acquire = [&](auto Pc) -> const float* {
            constexpr int P = decltype(Pc)::value;
            lsnf_panel_barrier();
            float* nxt = buf0 + ((P + 1) & 1) * C::SLOT;
            if constexpr (P + 1 < C::NP) {
                lsnf_issue_panel<C::kt_of(P + 1)>(gblk + C::off_of(P + 1), nxt, wave, lane);
            } else {
                if (more) lsnf_issue_panel<C::kt_of(0)>(gblk + C::BLOCK_FLOATS, nxt, wave, lane);
            }
            return buf0 + (P & 1) * C::SLOT;
        };

        // ---- S1: v = Wa^T x + ca  (actnorm model.py:244,268 folded into the 1x1 conv :187) ----
        f32x16 v[NZT];
        static_for<NZT>([&](auto nt) {
            const float* lb = acquire(std::integral_constant<int, nt.value>{});
            v[nt] = lsnf_bias_init(cb + 32 * nt, h);
            lsnf_panel_mma<C::KT1>(v[nt], x, lb, lane);
        });
        // logdet += sum(3*logs) (model.py:273-276); logdet += log|det W| (model.py:182,189)
        ell = ell + cb[32 * C::NP + 0];
        ell = ell + cb[32 * C::NP + 1];

        // ---- S2: h1 = relu(actnorm(v1 @ W1))  (model.py:326-328,307) ----
        f32x16 h1[WT];
        static_for<WT>([&](auto nt) {
            const float* lb = acquire(std::integral_constant<int, C::P1 + nt.value>{});
            h1[nt] = lsnf_bias_init(cb + 32 * (C::P1 + nt), h);
            lsnf_panel_mma<C::KT2>(h1[nt], v, lb, lane);
            h1[nt] = lsnf_relu16(h1[nt]);
        });
        // ---- S3: h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,308) ----
        f32x16 h2[WT];
        static_for<WT>([&](auto nt) {
            const float* lb = acquire(std::integral_constant<int, C::P1 + C::P2 + nt.value>{});
            h2[nt] = lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + nt), h);
            lsnf_panel_mma<C::KT3>(h2[nt], h1, lb, lane);
            h2[nt] = lsnf_relu16(h2[nt]);
        });
        // ---- S4: shift t / pre-sigmoid p = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) ----
        f32x16 tp[2 * HT];
        static_for<2 * HT>([&](auto nt) {
            const float* lb = acquire(std::integral_constant<int, C::P1 + C::P2 + C::P3 + nt.value>{});
            tp[nt] = lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + nt), h);
            lsnf_panel_mma<C::KT4>(tp[nt], h2, lb, lane);
        });
        // ---- coupling + per-sample log-scale reduction (model.py:414-418), concat (:422) ----
        float lsum = 0.0f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            x[t] = v[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(tp[HT + t][r], sig, lsig);
                x[HT + t][r] = (v[HT + t][r] + tp[t][r]) * sig;
                lsum += lsig;
            }
        }
        ell = ell + lsnf_pair_sum(lsum);

        if (a.z_saved != nullptr && more && live)
            lsnf_store_rows<HT>(x, a.z_saved + (size_t)blk * a.B * a.nz, sample, a.nz, a.half, h, a.vec4 != 0);
    }

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
    float ss = 0.0f;
#pragma unroll
    for (int t = 0; t < NZT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += x[t][r] * x[t][r];
    ss = lsnf_pair_sum(ss);
#ifdef LSNF_ABLATE_IO
    if (ss != 123.456f) return;   // keeps the computation alive, stores (practically) never happen
#endif
    if (live) {
        lsnf_store_rows<HT>(x, a.z_out, sample, a.nz, a.half, h, a.vec4 != 0);
        if (h == 0) {
            a.logdet_out[sample] = ell;
            if (a.ll_out) a.ll_out[sample] = (-0.5f * ss + 1.8378770664093453f) + ell;
        }
    }
}

template <class C>
hipError_t launch_fwd(const FwdArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.n_blocks * C::CONST_FLOATS + 2 * (size_t)C::SLOT) * sizeof(float);
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lsnf_fwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((a.B + LSNF_WG_SAMPLES - 1) / LSNF_WG_SAMPLES);
    hipLaunchKernelGGL(lsnf_fwd_kernel<C>, dim3(grid), dim3(LSNF_WG_THREADS), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

// host-side dispatcher (called from lsnf_api.hip)
hipError_t lsnf_launch_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                               const float* z_in, const float* objective, float* z_out, float* logdet_out,
                               float* ll_out, float* z_saved, int vec4, hipStream_t stream) {
    FwdArgs a;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels = plan + g.off_fwd_panels + (size_t)first_block * g.fwd_block_floats;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.z_saved = z_saved; a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_fwd<FwdCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_fwd<FwdCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_fwd<FwdCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
