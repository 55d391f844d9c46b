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
// panel pairs (two n-tiles x all k-tiles, <= 32 KiB, two independent accumulator chains), double-buffered,
// one barrier per pair.
#include <stdlib.h>
#include "lsnf_device.h"

namespace {

template <int HT_, int WT_>
struct FwdCfg : LsnfStackCfg<HT_, WT_> {
    using B = LsnfStackCfg<HT_, WT_>;
    static constexpr int BLOCK_FLOATS = B::FWD_BLOCK, CONST_FLOATS = B::FWD_CONST;
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
    float* act_saved;                  // NULL or the activation stash (lsnf_layout.h: LsnfActLayout) read by the SAVED backward
    int B, nz, half, n_blocks, vec4;
    double* stats;                     // NULL or 8 doubles: see lsnf_forward (in-kernel sum of ll / logdet over the batch)
    unsigned long long* stamps;        // LSNF_STAMPS diagnostic build only: [grid][4 waves][64] shader-clock stamps
};

#ifdef LSNF_STAMPS
#define LSNF_STAMP(i)                                                                                   \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * FWD_WAVES + wave) & 2047) * 64 + (i)] = t_; } while (0)
#define LSNF_STAMP_RT(i)                                                                                \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * FWD_WAVES + wave) & 2047) * 64 + (i)] = t_; } while (0)
#else
#define LSNF_STAMP(i) do {} while (0)
#define LSNF_STAMP_RT(i) do {} while (0)
#endif

// FWD_WAVES waves per workgroup, 2 waves per SIMD either way: 4 -> two workgroups of 128 rows per CU, 8 -> one workgroup of
// 256 rows per CU.  The 8-wave form halves the LDS-DMA pieces each wave has to issue per row (every panel is shared by
// twice as many rows; measured -5 us at B = 65536) but needs B > 32768 to put a workgroup on every CU.
template <class C, int FWD_WAVES>
__global__ __launch_bounds__(64 * FWD_WAVES, 2) void lsnf_fwd_kernel(const FwdArgs a) {
    constexpr int FWD_THREADS = 64 * FWD_WAVES;
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                         // n_blocks * CONST_FLOATS
    float* buf0 = smem + a.n_blocks * C::CONST_FLOATS;         // 2 x SLOT
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 31, h = lane >> 5;

    LSNF_STAMP(0);
    LSNF_STAMP_RT(50);
    // prologue: first panel pair in flight, constants to LDS, latent rows to registers
    LsnfPipeT<FWD_WAVES> pipe;
    pipe.buf0 = buf0; pipe.slot = C::SLOT; pipe.wave = wave; pipe.lane = lane;
    pipe.template prime<lsnf_first_ktc(C::P1, C::KT1)>(a.panels);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += FWD_THREADS) cst[i] = a.consts[i];

    const long sample = ((long)blockIdx.x * FWD_WAVES + wave) * 32 + m;
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
    lsnf_load_rows<HT>(x, a.z_in, row, a.nz, a.half, h, a.vec4);
    float ell = a.objective ? a.objective[row] : 0.0f;
#endif

    LSNF_STAMP(1);
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const size_t wtile = (size_t)blockIdx.x * FWD_WAVES + wave;          // this wave's 32-sample tile
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        float* act = (a.act_saved && wtile * 32 < (size_t)a.B)      // waves past the batch own no stash tile
                         ? a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile : nullptr;
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels + (size_t)blk * C::BLOCK_FLOATS;
        const bool more = blk + 1 < a.n_blocks;

        const float* gnext = more ? gblk + C::BLOCK_FLOATS : nullptr;
        auto keep = [](f32x16 acc, int) { return acc; };
        auto relu = [](f32x16 acc, int) { return lsnf_relu16(acc); };

        // ---- S1: v = Wa^T x + ca  (actnorm model.py:244,268 folded into the 1x1 conv :187) ----
        f32x16 v[NZT];
        lsnf_gemm_stage<C::P1, C::KT1, lsnf_first_ktc(C::P2, C::KT2)>(
            pipe, gblk, gblk + C::OFF_S2, v, x, [&](int t) { return lsnf_bias_init(cb + 32 * t, h); }, keep);
        LSNF_STAMP(2 + 6 * blk + 0);
        // last block: its v1 half is already final (the coupling passes it through, model.py:422) -- store it now, under
        // the MFMAs of S2..S4, instead of in the store burst at the end of the kernel
        if (!more && live) {
            float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
            for (int t = 0; t < HT; ++t) lsnf_store_tile<HT>(t, v[t], zo, a.half, h, a.vec4);
        }
        // logdet += sum(3*logs) (model.py:273-276); logdet += log|det W| (model.py:182,189)
        ell = ell + cb[32 * C::NP + 0];
        ell = ell + cb[32 * C::NP + 1];
        // ---- S2: h1 = relu(actnorm(v1 @ W1))  (model.py:326-328,307) ----
        f32x16 h1[WT];
        lsnf_gemm_stage<C::P2, C::KT2, lsnf_first_ktc(C::P3, C::KT3)>(
            pipe, gblk + C::OFF_S2, gblk + C::OFF_S3, h1, v, [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + t), h); }, relu);
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) *lsnf_act_mask_ptr(act, al.mask_off, t, lane) = lsnf_posmask16(h1[t]);
        }
        LSNF_STAMP(2 + 6 * blk + 1);
        // ---- S3: h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,308) ----
        f32x16 h2[WT];
        lsnf_gemm_stage<C::P3, C::KT3, lsnf_first_ktc(C::P4, C::KT4)>(
            pipe, gblk + C::OFF_S3, gblk + C::OFF_S4, h2, h1,
            [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + t), h); }, relu);
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) *lsnf_act_mask_ptr(act, al.mask_off, WT + t, lane) = lsnf_posmask16(h2[t]);
        }
        LSNF_STAMP(2 + 6 * blk + 2);
        // ---- S4: shift t / pre-sigmoid p = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) ----
        f32x16 tp[2 * HT];
        lsnf_gemm_stage<C::P4, C::KT4, lsnf_first_ktc(C::P1, C::KT1)>(
            pipe, gblk + C::OFF_S4, gnext, tp, h2,
            [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + t), h); }, keep);
        LSNF_STAMP(2 + 6 * blk + 3);
        // ---- coupling + per-sample log-scale reduction (model.py:414-418), concat (:422) ----
        float lsum = 0.0f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            x[t] = v[t];
            f32x16 sg;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, l2;
                lsnf_sigmoid_log2(tp[HT + t][r], sig, l2);
                x[HT + t][r] = (v[HT + t][r] + tp[t][r]) * sig;
                sg[r] = sig;
                lsum += l2;                                   // -log2(scale); one multiply by -ln2 per sample below
            }
            if (act) lsnf_act_store_sigma(act, t, sg, lane);
        }
        ell = ell + -0.6931471805599453f * lsnf_pair_sum(lsum);
        LSNF_STAMP(2 + 6 * blk + 4);

        if (a.z_saved != nullptr && more && live)
            lsnf_store_rows<HT>(x, a.z_saved + (size_t)blk * a.B * a.nz, sample, a.nz, a.half, h, a.vec4);
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
        {   // second half only: the first half went out during the last block
            float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
            for (int t = HT; t < NZT; ++t) lsnf_store_tile<HT>(t, x[t], zo, a.half, h, a.vec4);
        }
        if (h == 0) {
            a.logdet_out[sample] = ell;
            if (a.ll_out) a.ll_out[sample] = (-0.5f * ss + 1.8378770664093453f) + ell;
        }
    }
    if (a.stats) {   // kernel-uniform: batch sums of ll and logdet, one pair of fp64 atomics per workgroup
        float sl = (live && h == 0) ? ((-0.5f * ss + 1.8378770664093453f) + ell) : 0.0f;
        float sd = (live && h == 0) ? ell : 0.0f;
        double dl = (double)sl, dd = (double)sd;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
        __syncthreads();                               // the weight buffers are dead: reuse their first bytes
        double* red = reinterpret_cast<double*>(buf0);
        if (lane == 0) { red[2 * wave] = dl; red[2 * wave + 1] = dd; }
        __syncthreads();
        if (wave == 0) {         // (all 64 lanes: lsnf_publish_stats is a wave-level protocol)
            double tl = 0.0, td = 0.0;
            for (int w = 0; w < FWD_WAVES; ++w) { tl += red[2 * w]; td += red[2 * w + 1]; }
            lsnf_publish_stats(a.stats, tl, td, a.B, lane);
        }
    }
    LSNF_STAMP(40);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSNF_STAMP(41);
    LSNF_STAMP_RT(51);
}

template <class C, int FWD_WAVES>
hipError_t launch_fwd_w(const FwdArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.n_blocks * C::CONST_FLOATS + 2 * (size_t)C::SLOT) * sizeof(float);
    auto kern = lsnf_fwd_kernel<C, FWD_WAVES>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * FWD_WAVES - 1) / (32 * FWD_WAVES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FWD_WAVES), lds, stream, a);
    return hipGetLastError();
}

template <class C>
hipError_t launch_fwd(const FwdArgs& a, hipStream_t stream) {
    // more than 256 four-wave workgroups would double up on CUs anyway: switch to one eight-wave workgroup per CU
    static int forced = -1;                       // LSNF_FWD_WAVES=4|8: developer override for A/B runs
    if (forced < 0) { const char* e = getenv("LSNF_FWD_WAVES"); forced = e ? atoi(e) : 0; }
    const bool wide = forced ? forced == 8 : a.B > 256 * 128;
    return wide ? launch_fwd_w<C, 8>(a, stream) : launch_fwd_w<C, 4>(a, stream);
}
}  // namespace

#ifdef LSNF_STAMPS
unsigned long long* g_lsnf_stamps = nullptr;
extern "C" unsigned long long* lsnf_debug_stamps(void) { return g_lsnf_stamps; }
#endif
// host-side dispatcher (called from lsnf_api.hip)
hipError_t lsnf_launch_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                               const float* z_in, const float* objective, float* z_out, float* logdet_out,
                               float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                               hipStream_t stream) {
    FwdArgs a;
    a.stats = stats;
    a.act_saved = act_saved ? act_saved + (size_t)first_block * lsnf_act_layout(B, g.HT, g.WT).per_block : nullptr;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels = plan + g.off_fwd_panels + (size_t)first_block * g.fwd_block_floats;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.z_saved = z_saved; a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4;
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    {   // diagnostic build: a leaked device buffer, address published through LSNF_STAMPS_PTR (see tools/stamps.py)
        static unsigned long long* buf = nullptr;
        if (!buf) { if (hipMalloc(&buf, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) buf = nullptr; }
        a.stamps = (B <= 128 * 4096) ? buf : nullptr;
        extern unsigned long long* g_lsnf_stamps; g_lsnf_stamps = buf;
    }
#endif
    if (g.HT == 1 && g.WT == 1) return launch_fwd<FwdCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_fwd<FwdCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_fwd<FwdCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
