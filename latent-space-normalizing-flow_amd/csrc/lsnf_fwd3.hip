// lsnf_fwd3.hip -- the fused forward of lsnf_fwd.hip with its GEMMs on the bf16 matrix pipe at fp32 accuracy.
//
// Same math, same interface and same register layout as lsnf_fwd.hip (replaces reference model.py:473-483 +
// train.py:317-319); what changes is how a product  acc += W^T x  is evaluated.  fp32 MFMA
// (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s on MI355X, the bf16 MFMA (v_mfma_f32_32x32x16_bf16) at 2.5 PFLOP/s,
// 16x.  Both operands are split error-free into three bf16 terms,
//       w = w1 + w2 + w3,   x = x1 + x2 + x3      (each term the round-to-nearest bf16 of what is left),
// and the product keeps the six terms of weight >= 2^-18:
//       w x ~= w1 x1 + w1 x2 + w2 x1 + w2 x2 + w1 x3 + w3 x1,
// every bf16 x bf16 product being exact in the MFMA's fp32 accumulator.  The dropped terms are <= 2^-26 |w||x|,
// below the 2^-24 rounding of an fp32 product: the log-prob error against float64 is the same 2-3e-7 as the fp32
// MFMA path's (tools/study/split_bf16_accuracy.py, tests/test_gpu_forward.py) -- this is NOT a reduced-precision
// mode.  Six MFMAs of K = 16 replace eight of K = 2 x 8: 2.6x the fp32 MFMA rate (tools/micro/mfma_bf16.hip);
// the weights are split once in lsnf_prepare (plan region off_f3_panels, 6 KiB per 32x32 block instead of 4),
// the activations on the fly (4.5 VALU per element: v_cvt_pk_bf16_f32, shift/and, v_pk_add_f32).
//
// Work decomposition: one workgroup per CU (the two 48 KiB weight buffers do not fit twice) of 8 waves (batches above
// 32 768 rows) or 4 waves, wave = 32 samples; weights stream L2 -> LDS by LDS-DMA in panel pairs exactly as in lsnf_fwd.hip.  Because the k-slot j of
// lane-half h in k-step s is accumulator register 8*s + j (lsnf_layout.h), an output tile converted to bf16 pairs in
// register order is directly the next GEMM's B operand.
#include <stdlib.h>
#include "lsnf_l16.h"

namespace {



template <int HT_, int WT_>
struct Fwd3Cfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = L16_FRAG_FLOATS;
    static constexpr int OFF3_S2 = F * S::P1 * S::KT1;
    static constexpr int OFF3_S3 = OFF3_S2 + F * S::P2 * S::KT2;
    static constexpr int OFF3_S4 = OFF3_S3 + F * S::P3 * S::KT3;
    static constexpr int BLOCK3 = OFF3_S4 + F * S::P4 * S::KT4;
    static constexpr int SLOT3 = 2 * S::MAXKT * F;            // LDS floats of one panel pair
    static constexpr int CONST_FLOATS = S::FWD_CONST;
};

struct Fwd3Args {
    const float* consts; const float* panels3;
    const float* z_in; const float* objective;
    float* z_out; float* logdet_out; float* ll_out; float* z_saved; float* act_saved;
    int B, nz, half, n_blocks, vec4;
    double* stats;
    unsigned long long* stamps;        // LSNF_STAMPS diagnostic build only: [waves & 2047][64] clock stamps
    int shape16;                       // 1: the v_mfma_f32_16x16x32_bf16 variant (panels3 then points at its operand order)
    const unsigned* guard;             // fp16x2 kernel: the plan's guard words ([0]: weights outside fp16's range), read only
    int hdump_tiled;                   // h1 / h2 in the tiled form (lsnf_l16.h l16_store_tiled) instead of row-major
    float* hdump; int width;           // L16 kernels: NULL, or the parameter-gradient dump (LsnfDumpLayout): h1, h2 of every block
    int fixup;                         // bf16x3 L16 kernel: 1 = run as the fix-up pass of an fp16x2 launch: a workgroup
                                       // recomputes its rows only if one of its waves left LSNF_F16_SENTINEL_BITS in
                                       // logdet_out (lsnf_layout.h); the flag travels in the launch's own output
};

#ifdef LSNF_STAMPS   // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (tools/stamps_fwd3.py)
#define F3_STAMP(i, INSN)                                                                               \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile(INSN " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");  \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * F3_WAVES + wave) & 2047) * 64 + (i)] = t_; } while (0)
#else
#define F3_STAMP(i, INSN) do {} while (0)
#endif

#if LSNF_L16_PARTS == 3 && defined(LSNF_EXPERIMENTAL_KERNELS)   // the 32x32x16 comparison kernel: bf16x3 split, research builds only

// registers 8*s .. 8*s+7 of an activation tile -> x1, x2, x3 of k-step s
__device__ __forceinline__ void split_kstep(const f32x16& x, int s, Split3& out) {
    u32x4 w[3];
#ifdef LSNF_ABLATE_SPLIT   // timing diagnostic only (wrong numbers): prices the VALU work of the operand split
#pragma unroll
    for (int q = 0; q < 4; ++q) { const unsigned u = __builtin_bit_cast(unsigned, x[8 * s + 2 * q]); w[0][q] = u; w[1][q] = u; w[2][q] = u; }
#pragma unroll
    for (int i = 0; i < 3; ++i) out.p[i] = __builtin_bit_cast(bf16x8, w[i]);
    return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = x[8 * s + 2 * q], b = x[8 * s + 2 * q + 1];
        const unsigned p1 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p1 << 16); b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned p2 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        w[0][q] = p1; w[1][q] = p2; w[2][q] = pk_bf16(a, b);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) out.p[i] = __builtin_bit_cast(bf16x8, w[i]);
}
template <int KT>
__device__ __forceinline__ void split_tiles(const f32x16* x, Split3* out) {   // out[2*KT]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { split_kstep(x[kt], 0, out[2 * kt]); split_kstep(x[kt], 1, out[2 * kt + 1]); }
}


// NTILES (1 or 2) n-tiles x KT k-tiles out of one LDS buffer: acc_t += W_t^T in.  Per k-step 3 fragment reads (the
// three weight parts) feed 6 MFMAs; the next k-step's reads -- across the tile boundary too, the second tile's panel
// follows the first one's -- are issued before the current MFMAs (second register set), so a pair pays one pipeline
// fill.  One dependent chain is enough: the 8-pass bf16 MFMA issues back to back on its own accumulator
// (tools/micro/mfma_bf16.hip).  Measured and rejected (A/B in one job, tools/ablate_fwd3.sh): reads two k-steps ahead
// (+2 us), bias loads before the acquire barrier (+1 us), two interleaved accumulator chains (+5 us, spills).
template <int KT, int NTILES>
__device__ __forceinline__ void panel_mma3(f32x16& acc0, f32x16& acc1, const Split3* in, const float* lbuf, int lane) {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(lbuf) + lane;
    constexpr int STEPS = 2 * KT * NTILES;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = wp[p * 64];
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int idx = 0; idx < STEPS; ++idx) {
        const int ks = idx % (2 * KT);
        bf16x8 na[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) na[p] = a[p];
        if (idx + 1 < STEPS) {
#pragma unroll
            for (int p = 0; p < 3; ++p) na[p] = wp[((idx + 1) * 3 + p) * 64];
        }
#ifdef LSNF_ABLATE_MFMA    // timing diagnostic only (wrong numbers): everything but the matrix pipe (one MFMA per k-step kept)
#define LSNF_F3_MMA(WI, XI) if (WI == 0 && XI == 0) { if (idx < 2 * KT) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], in[ks].p[0], acc0, 0, 0, 0); else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], in[ks].p[0], acc1, 0, 0, 0); }
#else
#define LSNF_F3_MMA(WI, XI)                                                                                 \
        if (idx < 2 * KT) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[WI], in[ks].p[XI], acc0, 0, 0, 0); \
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[WI], in[ks].p[XI], acc1, 0, 0, 0);
#endif
        LSNF_F3_TERMS(LSNF_F3_MMA)
#undef LSNF_F3_MMA
        if (idx + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#ifdef LSNF_ABLATE_MFMA
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#else
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = na[p];
    }
}

// one GEMM stage out[t] = post(init(t) + W_t^T in), streamed as panel pairs (cf. lsnf_gemm_stage)
template <int NT, int KT, int NEXT_KIB, class Pipe, class Init, class Post>
__device__ __forceinline__ void gemm_stage3(Pipe& pipe, const float* gsrc, const float* gnext, f32x16* out, const Split3* in,
                                            Init&& init, Post&& post) {
    constexpr int NSP = (NT + 1) / 2;
    lsnf_static_for<NSP>([&](auto qc) {
        constexpr int q = decltype(qc)::value, t0 = 2 * q, cnt = (NT - t0 >= 2) ? 2 : 1;
        const float* lb;
        if constexpr (q + 1 < NSP) {
            constexpr int cn = (NT - (t0 + 2) >= 2) ? 2 : 1;
            lb = pipe.template acquire<6 * KT * cn>(gsrc + (t0 + 2) * KT * LSNF_FRAG3_FLOATS);
        } else {
            lb = pipe.template acquire<NEXT_KIB>(gnext);
        }
        out[t0] = init(t0);
        if constexpr (cnt == 2) out[t0 + 1] = init(t0 + 1);
        if constexpr (cnt == 2) {
            panel_mma3<KT, 2>(out[t0], out[t0 + 1], in, lb, pipe.lane);
            out[t0 + 1] = post(out[t0 + 1], t0 + 1);
        } else {
            panel_mma3<KT, 1>(out[t0], out[t0], in, lb, pipe.lane);
        }
        out[t0] = post(out[t0], t0);
    });
}

// F3_WAVES waves per workgroup, one workgroup per CU either way (the weight buffers): 8 = two waves per SIMD, 256 rows;
// 4 = one wave per SIMD, 128 rows -- twice the workgroups for batches that would leave CUs idle with 256-row groups
template <class C, int F3_WAVES>
__global__ __launch_bounds__(64 * F3_WAVES, 1) void lsnf_fwd3_kernel(const Fwd3Args a) {
    constexpr int THREADS = 64 * F3_WAVES;
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                         // n_blocks * CONST_FLOATS
    float* buf0 = smem + a.n_blocks * C::CONST_FLOATS;         // 2 x SLOT3
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 31, h = lane >> 5;

    F3_STAMP(0, "s_memtime");
    F3_STAMP(50, "s_memrealtime");
    Pipe3<F3_WAVES> pipe;
    pipe.buf0 = buf0; pipe.slot = C::SLOT3; pipe.wave = wave; pipe.lane = lane;
    pipe.template prime<first_kib(C::P1, C::KT1)>(a.panels3);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += THREADS) cst[i] = a.consts[i];

    const long sample = ((long)blockIdx.x * F3_WAVES + wave) * 32 + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;

    f32x16 x[NZT];
#ifdef LSNF_ABLATE_IO    // timing diagnostic only: no HBM reads of z
#pragma unroll
    for (int t = 0; t < NZT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = 0.001f * (float)(lane + r + t);
    float ell = 0.0f;
#else
    lsnf_load_rows<HT>(x, a.z_in, row, a.nz, a.half, h, a.vec4);
    float ell = a.objective ? a.objective[row] : 0.0f;
#endif

    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const size_t wtile = (size_t)blockIdx.x * F3_WAVES + wave;
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        float* act = (a.act_saved && wtile * 32 < (size_t)a.B)
                         ? a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile : nullptr;
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels3 + (size_t)blk * C::BLOCK3;
        const bool more = blk + 1 < a.n_blocks;
        const float* gnext = more ? gblk + C::BLOCK3 : nullptr;
        auto keep = [](f32x16 acc, int) { return acc; };
        auto relu = [](f32x16 acc, int) { return lsnf_relu16(acc); };

        // ---- S1: v = Wa^T x + ca  (actnorm model.py:244,268 folded into the 1x1 conv :187) ----
        f32x16 v[NZT];
        {
            Split3 xs[2 * NZT];
            split_tiles<NZT>(x, xs);
            gemm_stage3<C::P1, C::KT1, first_kib(C::P2, C::KT2)>(
                pipe, gblk, gblk + C::OFF3_S2, v, xs, [&](int t) { return lsnf_bias_init(cb + 32 * t, h); }, keep);
        }
        if (!more && live) {     // last block: the v1 half is final (model.py:422) -- store it under the MFMAs of S2..S4
            float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
            for (int t = 0; t < HT; ++t) lsnf_store_tile<HT>(t, v[t], zo, a.half, h, a.vec4);
        }
        ell = ell + cb[32 * C::NP + 0];          // sum(3*logs)  (model.py:273-276)
        ell = ell + cb[32 * C::NP + 1];          // log|det W|   (model.py:182,189)
        // ---- S2: h1 = relu(actnorm(v1 @ W1))  (model.py:326-328,307) ----
        f32x16 h1[WT];
        {
            Split3 vs[2 * HT];
            split_tiles<HT>(v, vs);
            gemm_stage3<C::P2, C::KT2, first_kib(C::P3, C::KT3)>(
                pipe, gblk + C::OFF3_S2, gblk + C::OFF3_S3, h1, vs,
                [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + t), h); }, relu);
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) *lsnf_act_mask_ptr(act, al.mask_off, t, lane) = lsnf_posmask16(h1[t]);
        }
        // ---- S3: h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,308) ----
        f32x16 h2[WT];
        {
            Split3 hs[2 * WT];
            split_tiles<WT>(h1, hs);
            gemm_stage3<C::P3, C::KT3, first_kib(C::P4, C::KT4)>(
                pipe, gblk + C::OFF3_S3, gblk + C::OFF3_S4, h2, hs,
                [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + t), h); }, relu);
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) *lsnf_act_mask_ptr(act, al.mask_off, WT + t, lane) = lsnf_posmask16(h2[t]);
        }
        // ---- S4: shift t / pre-sigmoid p = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) ----
        f32x16 tp[2 * HT];
        {
            Split3 hs[2 * WT];
            split_tiles<WT>(h2, hs);
            gemm_stage3<C::P4, C::KT4, first_kib(C::P1, C::KT1)>(
                pipe, gblk + C::OFF3_S4, gnext, tp, hs,
                [&](int t) { return lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + t), h); }, keep);
        }
        // ---- coupling + per-sample log-scale reduction (model.py:414-418), concat (:422) ----
        float lsum = 0.0f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            x[t] = v[t];
            f32x16 sg;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, l2;
#ifdef LSNF_ABLATE_EPILOGUE
                sig = tp[HT + t][r] * 0.25f + 0.5f; l2 = tp[HT + t][r];
#else
                lsnf_sigmoid_log2(tp[HT + t][r], sig, l2);
#endif
                x[HT + t][r] = (v[HT + t][r] + tp[t][r]) * sig;
                sg[r] = sig;
                lsum += l2;
            }
            if (act) lsnf_act_store_sigma(act, t, sg, lane);
        }
        ell = ell + -0.6931471805599453f * lsnf_pair_sum(lsum);
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
        float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
        for (int t = HT; t < NZT; ++t) lsnf_store_tile<HT>(t, x[t], zo, a.half, h, a.vec4);
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
        __syncthreads();
        double* red = reinterpret_cast<double*>(buf0);
        if (lane == 0) { red[2 * wave] = dl; red[2 * wave + 1] = dd; }
        __syncthreads();
        if (wave == 0) {         // (all 64 lanes: lsnf_publish_stats is a wave-level protocol)
            double tl = 0.0, td = 0.0;
            for (int w = 0; w < F3_WAVES; ++w) { tl += red[2 * w]; td += red[2 * w + 1]; }
            lsnf_publish_stats(a.stats, tl, td, a.B, lane);
        }
    }
    F3_STAMP(41, "s_memtime");
    F3_STAMP(51, "s_memrealtime");
}

#endif  // LSNF_L16_PARTS == 3 && LSNF_EXPERIMENTAL_KERNELS

// =====================================================================================================================
// The same kernel on v_mfma_f32_16x16x32_bf16 ("L16" lane layout of lsnf_layout.h: a wave's 32 samples are two sample
// tiles st of 16; lane = (n = lane & 15, g = lane >> 4); register (2*ft + st)*4 + r of a 32-feature activation tile
// holds feature 16*ft + 4*g + r of sample 16*st + n).  Same flops, same LDS traffic, same register count -- but on real
// data the chip sustains a higher clock under this MFMA shape (tools/micro/mfma_bf16_shapes.hip: +12 %).
// =====================================================================================================================
template <class C, int F3_WAVES>
__global__ __launch_bounds__(64 * F3_WAVES, 1) void lsnf_fwd3b_kernel(const Fwd3Args a) {
    constexpr int THREADS = 64 * F3_WAVES;
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;
    float* buf0 = smem + a.n_blocks * C::CONST_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n = lane & 15, g = lane >> 4;

    F3_STAMP(0, "s_memtime");
    F3_STAMP(50, "s_memrealtime");
    const int wbase = (blockIdx.x * F3_WAVES + wave) * 32;            // first row of this wave (B <= 2^28: 32-bit row indices)
#if LSNF_L16_PARTS == 3
    if (a.fixup) {                                // fix-up pass of the fp16 forward: only workgroups in which a wave raised its flag
        unsigned* fl = reinterpret_cast<unsigned*>(smem);
        unsigned f = 0u;
        if (wbase < a.B) f = __builtin_bit_cast(unsigned, a.logdet_out[(long)wbase]) == LSNF_F16_SENTINEL_BITS ? 1u : 0u;
        if (lane == 0) fl[wave] = f;
        __syncthreads();
        unsigned any = 0u;
#pragma unroll
        for (int w = 0; w < F3_WAVES; ++w) any |= fl[w];
        __syncthreads();                          // (smem is reused below)
        if (any == 0u) return;                    // workgroup-uniform
    }
#else
    if (a.guard[0] != 0u) {                       // weights outside fp16's range (prepare): leave every row to the fix-up pass
        if (lane == 0 && wbase < a.B) a.logdet_out[wbase] = __builtin_bit_cast(float, LSNF_F16_SENTINEL_BITS);
        return;
    }
    // Range guard.  An operand x with |x| >= 65520 splits into x1 = +-inf, x2 = x - x1 = -+inf, and every output of the
    // GEMM that consumes it becomes NaN (w*inf - w*inf, or 0*inf where w = 0): looking at ONE output register per
    // sample tile after each stage sees it -- before the ReLU, which would swallow the NaN (v_max_f32(NaN, 0) = 0).
    bool bad = false;
#endif
    Pipe3<F3_WAVES> pipe;
    pipe.buf0 = buf0; pipe.slot = C::SLOT3; pipe.wave = wave; pipe.lane = lane;
    pipe.template prime<first_kib(C::P1, C::KT1)>(a.panels3);

    int sample[2], rows[2]; bool live[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) { sample[st] = wbase + 16 * st + n; live[st] = sample[st] < a.B; rows[st] = live[st] ? sample[st] : a.B - 1; }

    f32x16 x[NZT];
#pragma unroll
    for (int t = 0; t < NZT; ++t) x[t] = l16_load_tile<HT>(t, a.z_in, rows, a.nz, a.half, g, a.vec4);
    __builtin_amdgcn_sched_barrier(0);
    // (the constant blocks are copied AFTER the row loads have gone out: the copy waits for its loads in order, one memory
    //  round trip that the rows would otherwise start behind)
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += THREADS) cst[i] = a.consts[i];
    float ell[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) ell[st] = a.objective ? a.objective[rows[st]] : 0.0f;

    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const bool w4 = (a.width & 3) == 0;
    const size_t wtile = (size_t)blockIdx.x * F3_WAVES + wave;
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        float* act = (a.act_saved && wtile * 32 < (size_t)a.B)
                         ? a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile : nullptr;
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels3 + (size_t)blk * C::BLOCK3;
        const bool more = blk + 1 < a.n_blocks;
        const float* gnext = more ? gblk + C::BLOCK3 : nullptr;
#if LSNF_L16_PARTS == 3
        auto keep = [](f32x16 acc, int) { return acc; };
        auto relu = [](f32x16 acc, int) { return lsnf_relu16(acc); };
#else
        auto keep = [&](f32x16 acc, int t) { if (t == 0) bad = bad || acc[0] != acc[0] || acc[4] != acc[4]; return acc; };
        auto relu = [&](f32x16 acc, int t) { if (t == 0) bad = bad || acc[0] != acc[0] || acc[4] != acc[4]; return lsnf_relu16(acc); };
#endif

        // ---- S1: v = Wa^T x + ca  (model.py:244,268,187) ----
        f32x16 v[NZT];
        {
            Split3 xs[2 * NZT];
            l16_split_tiles<NZT>(x, xs);
            l16_gemm_stage3<C::P1, C::KT1, first_kib(C::P2, C::KT2)>(
                pipe, gblk, gblk + C::OFF3_S2, v, xs, [&](int t) { return l16_bias_init(cb + 32 * t, g); }, keep);
        }
        // f_width 128 with two waves per SIMD (256 registers): the four split tiles of h1 / h2 (96 registers), four finished and two
        // running accumulators of S3 / S4 do not fit beside v (64): hipcc spills ~49 registers to scratch there.  Tried instead
        // (PARK, off): v1 / v2 parked in this wave's OWN z_out rows between S2's split and the end of the block -- zero scratch,
        // but 262 MB of extra row traffic per launch at B = 65 536: 186 us against 153 us with hipcc's spills (bf16x3; fp16x2
        // 133 vs 92; tools/ab_c5.py, two alternations on one box).  The compiler's choice of WHAT to spill is the cheaper one.
#ifdef LSNF_PARK_V
        constexpr bool PARK = WT == 4 && F3_WAVES == 8;
#else
        constexpr bool PARK = false;
#endif
        if (PARK || !more) {     // last block: the v1 half is final (model.py:422) -- its stores drain under the MFMAs of S2..S4
#pragma unroll
            for (int t = 0; t < HT; ++t) l16_store_tile<HT>(t, v[t], a.z_out, sample, live, a.nz, a.half, g, a.vec4);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) { ell[st] = ell[st] + cb[32 * C::NP + 0]; ell[st] = ell[st] + cb[32 * C::NP + 1]; }
        if constexpr (PARK) {
#pragma unroll
            for (int t = HT; t < NZT; ++t) l16_store_tile<HT>(t, v[t], a.z_out, sample, live, a.nz, a.half, g, a.vec4);
        }
        // ---- S2: h1 = relu(actnorm(v1 @ W1))  (model.py:326-328,307) ----
        f32x16 h1[WT];
        {
            Split3 vs[2 * HT];
            l16_split_tiles<HT>(v, vs);
            l16_gemm_stage3<C::P2, C::KT2, first_kib(C::P3, C::KT3)>(
                pipe, gblk + C::OFF3_S2, gblk + C::OFF3_S3, h1, vs,
                [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + t), g); }, relu);
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) l16_store_masks(reinterpret_cast<unsigned*>(act + al.mask_off) + t * 64, h1[t], n, g);
        }
        if (a.hdump) {           // kernel-uniform: h1 for the batch contraction dW2' = h1^T g_a2 (lsnf_params.hip)
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if (a.hdump_tiled) { if (wtile * 32 < (size_t)a.B) l16_store_tiled(h1[t], a.hdump + (size_t)blk * dl.per_block + dl.off_h1, wtile, a.width, t, n, g); }
                else l16_store_plain(h1[t], a.hdump + (size_t)blk * dl.per_block + dl.off_h1, sample, live, a.width, t, g, w4);
            }
        }
        // ---- S3: h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,308) ----
        f32x16 h2[WT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(h1, hs);
            l16_gemm_stage3<C::P3, C::KT3, first_kib(C::P4, C::KT4)>(
                pipe, gblk + C::OFF3_S3, gblk + C::OFF3_S4, h2, hs,
                [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + C::P2 + t), g); }, relu);
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < WT; ++t) l16_store_masks(reinterpret_cast<unsigned*>(act + al.mask_off) + (WT + t) * 64, h2[t], n, g);
        }
        if (a.hdump) {
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if (a.hdump_tiled) { if (wtile * 32 < (size_t)a.B) l16_store_tiled(h2[t], a.hdump + (size_t)blk * dl.per_block + dl.off_h2, wtile, a.width, t, n, g); }
                else l16_store_plain(h2[t], a.hdump + (size_t)blk * dl.per_block + dl.off_h2, sample, live, a.width, t, g, w4);
            }
        }
        // ---- S4: shift t / pre-sigmoid p = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) ----
        f32x16 tp[2 * HT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(h2, hs);
            l16_gemm_stage3<C::P4, C::KT4, first_kib(C::P1, C::KT1)>(
                pipe, gblk + C::OFF3_S4, gnext, tp, hs,
                [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + t), g); }, keep);
        }
        if constexpr (PARK) {        // v back from its parking place (rows of dead samples: the clamped row's values, never stored)
#pragma unroll
            for (int t = 0; t < NZT; ++t) v[t] = l16_load_tile<HT>(t, a.z_out, rows, a.nz, a.half, g, a.vec4);
        }
        // ---- coupling + per-sample log-scale reduction (model.py:414-418), concat (:422) ----
        float lsum[2] = {0.0f, 0.0f};
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
                lsum[(r >> 2) & 1] += l2;                      // register (2*ft + st)*4 + r': sample tile st = bit 2
            }
            if (act) l16_store_sigma(act, t, sg, n, g);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) ell[st] = ell[st] + -0.6931471805599453f * l16_group_sum(lsum[st]);
        if (a.z_saved != nullptr && more) {
#pragma unroll
            for (int t = 0; t < NZT; ++t)
                l16_store_tile<HT>(t, x[t], a.z_saved + (size_t)blk * a.B * a.nz, sample, live, a.nz, a.half, g, a.vec4);
        }
    }

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
    float ss[2] = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NZT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss[(r >> 2) & 1] += x[t][r] * x[t][r];
#pragma unroll
    for (int t = HT; t < NZT; ++t) l16_store_tile<HT>(t, x[t], a.z_out, sample, live, a.nz, a.half, g, a.vec4);
#if LSNF_L16_PARTS == 2
    // an operand at or beyond fp16's range (NaN compares false: it propagates by itself): this wave's results are not to
    // be trusted -- its first logdet element carries the flag, the bf16x3 fix-up pass queued behind recomputes the workgroup
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
#endif
    float ll[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        ll[st] = (-0.5f * l16_group_sum(ss[st]) + 1.8378770664093453f) + ell[st];
        if (live[st] && g == 0) {
            int smp = sample[st];
            asm volatile("" : "+v"(smp));       // (keeps hipcc from forming the two store addresses in the prologue and carrying
                                                //  them -- 4 registers -- through the whole stack: they were the last scratch spill)
            float ld = ell[st];
#if LSNF_L16_PARTS == 2
            if (wave_bad && st == 0 && n == 0) ld = __builtin_bit_cast(float, LSNF_F16_SENTINEL_BITS);
#endif
            a.logdet_out[smp] = ld;
            if (a.ll_out) a.ll_out[smp] = ll[st];
        }
    }
    if (a.stats) {   // kernel-uniform: batch sums of ll and logdet, one pair of fp64 atomics per workgroup
        double dl = 0.0, dd = 0.0;
#pragma unroll
        for (int st = 0; st < 2; ++st)
            if (live[st] && g == 0) { dl += (double)ll[st]; dd += (double)ell[st]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
        __syncthreads();
        double* red = reinterpret_cast<double*>(buf0);
        if (lane == 0) { red[2 * wave] = dl; red[2 * wave + 1] = dd; }
        __syncthreads();
        if (wave == 0) {         // (all 64 lanes: lsnf_publish_stats is a wave-level protocol)
            double tl = 0.0, td = 0.0;
            for (int w = 0; w < F3_WAVES; ++w) { tl += red[2 * w]; td += red[2 * w + 1]; }
            lsnf_publish_stats(a.stats, tl, td, a.B, lane);
        }
    }
    F3_STAMP(41, "s_memtime");
    F3_STAMP(51, "s_memrealtime");
}

template <class C, int F3_WAVES>
hipError_t launch_fwd3_w(const Fwd3Args& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.n_blocks * C::CONST_FLOATS + 2 * (size_t)C::SLOT3) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#if LSNF_L16_PARTS == 3 && defined(LSNF_EXPERIMENTAL_KERNELS)
    auto kern = a.shape16 ? lsnf_fwd3b_kernel<C, F3_WAVES> : lsnf_fwd3_kernel<C, F3_WAVES>;
#else
    auto kern = lsnf_fwd3b_kernel<C, F3_WAVES>;
#endif
    static unsigned long long lds_ok[2] = {0, 0};
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok[a.shape16 ? 1 : 0]); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * F3_WAVES - 1) / (32 * F3_WAVES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * F3_WAVES), lds, stream, a);
    return hipGetLastError();
}
template <class C>
hipError_t launch_fwd3(const Fwd3Args& a, hipStream_t stream) {
    // 256-row workgroups need B > 32768 to put one on (almost) every CU; below that 128-row workgroups use twice the CUs
    static const char* fw = getenv("LSNF_FORCE_WAVES");   // experiment knob (tools/): 4 or 8
    bool eight = fw ? atoi(fw) == 8 : a.B > 128 * 256;
    if (!a.shape16 && C::WT == 4) eight = false;          // (the 32x32x16 comparison kernel does not fit 256 registers at f_width 128)
    return eight ? launch_fwd3_w<C, 8>(a, stream) : launch_fwd3_w<C, 4>(a, stream);
}
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); hipErrorInvalidValue = this geometry is not covered
#ifndef LSNF_FWD3_ENTRY
#define LSNF_FWD3_ENTRY lsnf_launch_forward3
#endif
hipError_t LSNF_FWD3_ENTRY(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                int shape16, int fixup, hipStream_t stream, float* hdump, int hdump_tiled) {
    Fwd3Args a;
    a.shape16 = shape16;
    a.hdump_tiled = hdump_tiled;
    a.hdump = hdump ? hdump + (size_t)first_block * lsnf_dump_layout(B, g.nz, g.width).per_block : nullptr;
    a.width = g.width;
    if (hdump && !shape16) return hipErrorInvalidValue;      // (the dump is written by the L16 kernels)
    a.fixup = fixup;
    a.guard = reinterpret_cast<const unsigned*>(plan + g.off_guard);
#if !defined(LSNF_EXPERIMENTAL_KERNELS)
    if (!shape16) return hipErrorInvalidValue;               // the 32x32x16 comparison kernel is not in this build
#endif
#if LSNF_L16_PARTS == 3
    if (fixup && !shape16) return hipErrorInvalidValue;      // the fix-up pass is the L16 kernel
    if (fixup && stats) return hipErrorInvalidValue;         // (a partial recomputation cannot repair in-kernel batch sums)
#else
    if (stats) return hipErrorInvalidValue;                  // the fp16 forward is not used with in-kernel batch sums
#endif
    a.stats = stats;
    a.act_saved = act_saved ? act_saved + (size_t)first_block * lsnf_act_layout(B, g.HT, g.WT).per_block : nullptr;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
#if LSNF_L16_PARTS == 3
    a.panels3 = plan + (shape16 ? g.off_f3b_panels : g.off_f3_panels) + (size_t)first_block * g.f3_block_floats;
#else
    a.panels3 = plan + g.off_f2h_panels + (size_t)first_block * g.f2h_block_floats;
#endif
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.z_saved = z_saved; a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4;
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    { extern unsigned long long* g_lsnf_stamps;
      if (!g_lsnf_stamps) { if (hipMalloc(&g_lsnf_stamps, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) g_lsnf_stamps = nullptr; }
      a.stamps = g_lsnf_stamps; }
#endif
    if (g.HT == 1 && g.WT == 1) return launch_fwd3<Fwd3Cfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_fwd3<Fwd3Cfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_fwd3<Fwd3Cfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
