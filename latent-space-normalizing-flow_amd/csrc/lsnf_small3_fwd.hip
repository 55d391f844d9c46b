// lsnf_small3_fwd.hip -- latency forward of the flow stack on the bf16 matrix pipe (error-free bf16x3 split, see
// lsnf_fwd3.hip) for small / medium batches: the reference's own batch size (B = 100).
//
// Same math, same prepared weights (the 16x16x32 operand-order panels, plan region off_f3b_panels) and same ABI call
// as every other forward (replaces reference model.py:473-483 + train.py:317-319).  v_mfma_f32_16x16x32_bf16 handles
// 16 samples natively, so a workgroup owns SIXTEEN samples (7 workgroups at B = 100 instead of 4) and its 4 waves split
// every GEMM stage by half-units of 16 output features x all k: no K-splitting, stage outputs are final, the PRODUCER
// applies the activation, splits the 4 values it holds per lane into their three bf16 terms (18 VALU per unit) and
// stores them in B-operand order -- consumers read ready operands (3 ds_read_b128 per 32-feature k-tile).  Wave w owns
// v1[w], v2[w], t[w], p[w], so the coupling (model.py:414-418) stays in registers; a block costs 4 barriers.
// MFMA chain per wave and block: 96 x 16 cycles instead of 128 x 64; the kernel is bound by streaming the weights
// (192 KiB per block) through the CU's vector-memory path.  Weights go L2 -> VGPR two stages ahead, never under a branch.
//
// Lane layout "L16" (lsnf_layout.h): lane = (n = lane & 15 -> sample, g = lane >> 4); a half-unit (16 features x 16
// samples) is 4 registers per lane: feature 16*ft + 4*g + r.
#include "lsnf_small3.h"

namespace {

template <int HT_, int WT_>
struct Small3Cfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = LSNF_FRAG3_FLOATS;
    static constexpr int OFF3_S2 = F * S::P1 * S::KT1;
    static constexpr int OFF3_S3 = OFF3_S2 + F * S::P2 * S::KT2;
    static constexpr int OFF3_S4 = OFF3_S3 + F * S::P3 * S::KT3;
    static constexpr int BLOCK3 = OFF3_S4 + F * S::P4 * S::KT4;
    static constexpr int CONST_FLOATS = S::FWD_CONST;
    static constexpr int NU2 = (2 * WT_ + 3) / 4;              // hidden half-units per wave
    // LDS map (floats): X ping-pong (2 x NZT B-tiles), H1, H2 (WT B-tiles each), mask words (2 x WT x 32), reductions
    static constexpr int L_X = 0;
    static constexpr int L_H1 = L_X + 2 * S::NZT * S3_BTILE_FLOATS;
    static constexpr int L_H2 = L_H1 + WT_ * S3_BTILE_FLOATS;
    static constexpr int L_MASK = L_H2 + WT_ * S3_BTILE_FLOATS;
    static constexpr int L_RED = L_MASK + 2 * WT_ * 32;
    static constexpr int L_CONST = L_RED + 4 * 16 * 2;
};

struct Small3Args {
    const float* consts; const float* panels3b;
    const float* z_in; const float* objective;
    float* z_out; float* logdet_out; float* ll_out; float* z_saved; float* act_saved;
    double* stats;
    float* hdump;                      // NULL, or the parameter-gradient dump (LsnfDumpLayout) of block first_block: h1, h2 are written
    int B, nz, half, n_blocks, vec4, width;
};

template <class C>
__global__ __launch_bounds__(256, 1) void lsnf_small3_fwd_kernel(const Small3Args a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XB = smem + C::L_X;
    float* H1B = smem + C::L_H1;
    float* H2B = smem + C::L_H2;
    unsigned* MASK = reinterpret_cast<unsigned*>(smem + C::L_MASK);      // [h1 | h2][WT][32]
    float* RED = smem + C::L_RED;
    float* cst = smem + C::L_CONST;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;

    // this wave's half-units: hu1 of the latent halves (v1 / v2 / t / p), hw[i] of the hidden layers
    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }

    // weights two stages ahead: S1 and S2 of the first block are in flight before anything else
    UFrags<NZT> w1a = fetch_unit<NZT>(a.panels3b, nt1, ft1, lane);
    UFrags<NZT> w1b = fetch_unit<NZT>(a.panels3b, HT + nt1, ft1, lane);
    UFrags<HT> w2[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) w2[i] = fetch_unit<HT>(a.panels3b + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);

    const long sample = (long)blockIdx.x * S3_SAMPLES + n;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    // prologue: z rows -> B-operand tiles of block 0's input.  The row loads go out BEFORE the constant blocks are copied:
    // that copy travels through registers (load, wait, LDS store) and its wait is in order, so with the copy in front the
    // rows were requested one memory round trip later than necessary
    constexpr int NHU = (2 * NZT + 3) / 4;
    f32x4 xrow[NHU];
#pragma unroll
    for (int j = 0; j < NHU; ++j) {
        const int hu = wave + 4 * j;
        xrow[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hu < 2 * NZT) xrow[j] = load_row_half<HT>(hu >> 1, hu & 1, a.z_in + row * (long)a.nz, a.half, g, a.vec4);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += 256) cst[i] = a.consts[i];
    for (int i = tid; i < 2 * WT * 32; i += 256) MASK[i] = 0u;
#pragma unroll
    for (int j = 0; j < NHU; ++j) {
        const int hu = wave + 4 * j;
        if (hu < 2 * NZT) store_half(XB + (hu >> 1) * S3_BTILE_FLOATS, hu & 1, xrow[j], lane);
    }
    float ell = (wave == 0 && a.objective) ? a.objective[row] : 0.0f;      // per-wave partial of the running log-det
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const bool w4 = (a.width & 3) == 0;
    const size_t wtile = (size_t)(blockIdx.x >> 1);                       // 32-sample stash tile this workgroup is one half of
    const int st = (int)((blockIdx.x) & 1);                               // which half of that tile this workgroup is
    const int lane32 = 16 * st + n + 32 * (g & 1);                        // stash lane of (sample, feature-group parity)
    __syncthreads();

    f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, y2 = {0.f, 0.f, 0.f, 0.f};
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels3b + (size_t)blk * C::BLOCK3;
        const bool more = blk + 1 < a.n_blocks;
        const float* gnext = more ? gblk + C::BLOCK3 : gblk;             // last block re-fetches its own panels: no loads under a branch
        float* Xc = XB + (blk & 1) * NZT * S3_BTILE_FLOATS;
        float* Xn = XB + ((blk + 1) & 1) * NZT * S3_BTILE_FLOATS;
        float* act = a.act_saved ? a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile : nullptr;
        float* hd = (a.hdump && live) ? a.hdump + (size_t)blk * dl.per_block + sample * (long)a.width : nullptr;   // this sample's row of h1 / h2

        // ---- S1: v = Wa^T x + ca (model.py:244,268,187); this wave: v1[hu1], v2[hu1] ----
        UFrags<WT> w3[NU2];
#pragma unroll
        for (int i = 0; i < NU2; ++i) w3[i] = fetch_unit<WT>(gblk + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
        v1 = unit_mma<NZT>(unit_bias(cb + 32 * nt1, ft1, g), w1a, Xc, lane);
        f32x4 v2 = unit_mma<NZT>(unit_bias(cb + 32 * (HT + nt1), ft1, g), w1b, Xc, lane);
        if (has1) store_half(Xn + nt1 * S3_BTILE_FLOATS, ft1, v1, lane);  // S2's input = first half of the next block's input
        if (wave == 0) { ell = ell + cb[32 * C::NP + 0]; ell = ell + cb[32 * C::NP + 1]; }   // model.py:273-276, 182,189
        __syncthreads();
        // ---- S2: h1 = relu(actnorm(v1 @ W1)) (model.py:326-328,307) ----
        UFrags<WT> w4t = fetch_unit<WT>(gblk + C::OFF3_S4, nt1, ft1, lane);
        UFrags<WT> w4p = fetch_unit<WT>(gblk + C::OFF3_S4, HT + nt1, ft1, lane);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            const f32x4 h = relu4(unit_mma<HT>(unit_bias(cb + 32 * (C::P1 + nt), ft, g), w2[i], Xn, lane));
            if (hasw[i]) {
                store_half(H1B + nt * S3_BTILE_FLOATS, ft, h, lane);
                if (hd) store_plain_half(h, hd + dl.off_h1, a.width, nt, ft, g, w4);
                if (act) {
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c |= (h[r] > 0.0f ? 1u : 0u) << (4 * (2 * ft + (g >> 1)) + r);
                    atomicOr(&MASK[nt * 32 + n + 16 * (g & 1)], c);
                }
            }
        }
        __syncthreads();
        // ---- S3: h2 = relu(actnorm(h1 @ W2)) (model.py:326-328,308) ----
        w1a = fetch_unit<NZT>(gnext, nt1, ft1, lane);
        w1b = fetch_unit<NZT>(gnext, HT + nt1, ft1, lane);
        if (act && tid < WT * 32) {      // h1's mask words are complete: out to the stash, slots re-armed
            const int t = tid >> 5, j = tid & 31;                          // j = n + 16*(g&1) of the word
            reinterpret_cast<unsigned*>(act + al.mask_off)[t * 64 + 16 * st + (j & 15) + 32 * (j >> 4)] = MASK[t * 32 + j];
            MASK[t * 32 + j] = 0u;
        }
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            const f32x4 h = relu4(unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + nt), ft, g), w3[i], H1B, lane));
            if (hasw[i]) {
                store_half(H2B + nt * S3_BTILE_FLOATS, ft, h, lane);
                if (hd) store_plain_half(h, hd + dl.off_h2, a.width, nt, ft, g, w4);
                if (act) {
                    unsigned c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c |= (h[r] > 0.0f ? 1u : 0u) << (4 * (2 * ft + (g >> 1)) + r);
                    atomicOr(&MASK[(WT + nt) * 32 + n + 16 * (g & 1)], c);
                }
            }
        }
        __syncthreads();
        // ---- S4: shift t / pre-sigmoid p (model.py:347-349,411-413) + coupling (:414-418), in registers ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) w2[i] = fetch_unit<HT>(gnext + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
        if (act && tid < WT * 32) {
            const int t = tid >> 5, j = tid & 31;
            reinterpret_cast<unsigned*>(act + al.mask_off)[(WT + t) * 64 + 16 * st + (j & 15) + 32 * (j >> 4)] = MASK[(WT + t) * 32 + j];
            MASK[(WT + t) * 32 + j] = 0u;
        }
        const f32x4 tt_ = unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + nt1), ft1, g), w4t, H2B, lane);
        const f32x4 pp = unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + HT + nt1), ft1, g), w4p, H2B, lane);
        float lsum = 0.0f;
        f32x4 sg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sig, l2;
            lsnf_sigmoid_log2(pp[r], sig, l2);
            y2[r] = (v2[r] + tt_[r]) * sig;
            sg[r] = sig;
            lsum += l2;
        }
        if (has1) {
            ell = ell + -0.6931471805599453f * group_sum(lsum);
            store_half(Xn + (HT + nt1) * S3_BTILE_FLOATS, ft1, y2, lane);
            if (act) reinterpret_cast<f32x4*>(act + (size_t)nt1 * 1024)[(2 * ft1 + (g >> 1)) * 64 + lane32] = sg;
            if (a.z_saved != nullptr && more && live) {
                float* zr = a.z_saved + ((size_t)blk * a.B + sample) * a.nz;
                store_row_half<HT>(nt1, ft1, v1, zr, a.half, g, a.vec4);
                store_row_half<HT>(HT + nt1, ft1, y2, zr, a.half, g, a.vec4);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
    float ss = 0.0f;
    if (has1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ss += v1[r] * v1[r] + y2[r] * y2[r];
        if (live) {
            float* zr = a.z_out + sample * (long)a.nz;
            store_row_half<HT>(nt1, ft1, v1, zr, a.half, g, a.vec4);
            store_row_half<HT>(HT + nt1, ft1, y2, zr, a.half, g, a.vec4);
        }
    }
    ss = group_sum(ss);
    if (g == 0) { RED[(wave * 16 + n) * 2] = ss; RED[(wave * 16 + n) * 2 + 1] = ell; }
    __syncthreads();
    if (wave == 0) {
        float s2 = 0.0f, el = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s2 += RED[(w * 16 + n) * 2]; el += RED[(w * 16 + n) * 2 + 1]; }
        const float ll = (-0.5f * s2 + 1.8378770664093453f) + el;
        if (live && g == 0) {
            a.logdet_out[sample] = el;
            if (a.ll_out) a.ll_out[sample] = ll;
        }
        if (a.stats) {
            double dl = (live && g == 0) ? (double)ll : 0.0, dd = (live && g == 0) ? (double)el : 0.0;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
            lsnf_publish_stats(a.stats, dl, dd, a.B, lane);       // (wave-level protocol: all 64 lanes of wave 0)
        }
    }
}

// ---- restash: the activation stash of a forward that was run WITHOUT one, rebuilt from the block outputs ------------
// lsnf_backward_z without a stash used to recompute the coupling MLP inside the fp32-MFMA backward (54 us at B = 100).  The
// MLP's input is the first half of the block's OUTPUT (v1 passes through the coupling, model.py:422), so the stash --
// sigmoid(p) and the two ReLU masks -- follows from z_out / z_saved alone: S2, S3 and the pre-sigmoid half of S4 of every
// block, blocks independent of each other (grid.y).  Backward = this kernel + the from-the-stash backward, both on the
// bf16 matrix pipe.
struct RestashArgs {
    const float* consts; const float* panels3b;
    const float* z_out; const float* z_saved; float* act_saved;
    int B, nz, half, depth, vec4;
};
template <class C>
__global__ __launch_bounds__(256, 1) void lsnf_small3_restash_kernel(const RestashArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NU2 = C::NU2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* V1B = smem;                                          // HT B-tiles: v1
    float* H1B = V1B + HT * S3_BTILE_FLOATS;
    float* H2B = H1B + WT * S3_BTILE_FLOATS;
    unsigned* MASK = reinterpret_cast<unsigned*>(H2B + WT * S3_BTILE_FLOATS);      // [h1 | h2][WT][32]
    float* cst = reinterpret_cast<float*>(MASK + 2 * WT * 32);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int blk = blockIdx.y;
    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }
    const float* gblk = a.panels3b + (size_t)blk * C::BLOCK3;
    UFrags<HT> w2[NU2];
    UFrags<WT> w3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) {
        w2[i] = fetch_unit<HT>(gblk + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
        w3[i] = fetch_unit<WT>(gblk + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
    }
    const UFrags<WT> w4p = fetch_unit<WT>(gblk + C::OFF3_S4, HT + nt1, ft1, lane);
    const float* cb = a.consts + (size_t)blk * C::CONST_FLOATS;
    for (int i = tid; i < C::CONST_FLOATS; i += 256) cst[i] = cb[i];
    for (int i = tid; i < 2 * WT * 32; i += 256) MASK[i] = 0u;
    const long sample = (long)blockIdx.x * S3_SAMPLES + n;
    const long row = sample < a.B ? sample : (long)a.B - 1;
    const float* ysrc = (blk == a.depth - 1) ? a.z_out + row * (long)a.nz : a.z_saved + ((size_t)blk * a.B + row) * a.nz;
    for (int hu = wave; hu < 2 * HT; hu += 4)
        store_half(V1B + (hu >> 1) * S3_BTILE_FLOATS, hu & 1, load_row_half<HT>(hu >> 1, hu & 1, ysrc, a.half, g, a.vec4), lane);
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const size_t wtile = (size_t)(blockIdx.x >> 1);
    const int st = (int)(blockIdx.x & 1);
    const int lane32 = 16 * st + n + 32 * (g & 1);
    float* act = a.act_saved + (size_t)blk * al.per_block + wtile * al.per_tile;
    __syncthreads();
    // ---- S2 (model.py:326-328,307) ----
#pragma unroll
    for (int i = 0; i < NU2; ++i) {
        const int nt = hw[i] >> 1, ft = hw[i] & 1;
        const f32x4 h = relu4(unit_mma<HT>(unit_bias(cst + 32 * (C::P1 + nt), ft, g), w2[i], V1B, lane));
        if (hasw[i]) {
            store_half(H1B + nt * S3_BTILE_FLOATS, ft, h, lane);
            unsigned c = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) c |= (h[r] > 0.0f ? 1u : 0u) << (4 * (2 * ft + (g >> 1)) + r);
            atomicOr(&MASK[nt * 32 + n + 16 * (g & 1)], c);
        }
    }
    __syncthreads();
    // ---- S3 (model.py:326-328,308) ----
#pragma unroll
    for (int i = 0; i < NU2; ++i) {
        const int nt = hw[i] >> 1, ft = hw[i] & 1;
        const f32x4 h = relu4(unit_mma<WT>(unit_bias(cst + 32 * (C::P1 + C::P2 + nt), ft, g), w3[i], H1B, lane));
        if (hasw[i]) {
            store_half(H2B + nt * S3_BTILE_FLOATS, ft, h, lane);
            unsigned c = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) c |= (h[r] > 0.0f ? 1u : 0u) << (4 * (2 * ft + (g >> 1)) + r);
            atomicOr(&MASK[(WT + nt) * 32 + n + 16 * (g & 1)], c);
        }
    }
    __syncthreads();
    // ---- masks out; pre-sigmoid half of S4 (model.py:347-349,413) -> sigma ----
    if (tid < 2 * WT * 32) {
        const int t = tid >> 5, j = tid & 31;
        reinterpret_cast<unsigned*>(act + al.mask_off)[t * 64 + 16 * st + (j & 15) + 32 * (j >> 4)] = MASK[t * 32 + j];
    }
    const f32x4 pp = unit_mma<WT>(unit_bias(cst + 32 * (C::P1 + C::P2 + C::P3 + HT + nt1), ft1, g), w4p, H2B, lane);
    if (has1) {
        f32x4 sg;
#pragma unroll
        for (int r = 0; r < 4; ++r) { float sig, l2; lsnf_sigmoid_log2(pp[r], sig, l2); sg[r] = sig; }
        reinterpret_cast<f32x4*>(act + (size_t)nt1 * 1024)[(2 * ft1 + (g >> 1)) * 64 + lane32] = sg;
    }
}

template <class C>
hipError_t launch_small3_fwd(const Small3Args& a, hipStream_t stream) {
    const size_t lds = ((size_t)C::L_CONST + (size_t)a.n_blocks * C::CONST_FLOATS) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_small3_fwd_kernel<C>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + S3_SAMPLES - 1) / S3_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); hipErrorInvalidValue = this geometry is not covered
hipError_t lsnf_launch_small3_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                      const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                      float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                      hipStream_t stream, float* hdump) {
    Small3Args a;
    a.hdump = hdump ? hdump + (size_t)first_block * lsnf_dump_layout(B, g.nz, g.width).per_block : nullptr;
    a.width = g.width;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels3b = plan + g.off_f3b_panels + (size_t)first_block * g.f3_block_floats;
    a.act_saved = act_saved ? act_saved + (size_t)first_block * lsnf_act_layout(B, g.HT, g.WT).per_block : nullptr;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.z_saved = z_saved; a.stats = stats; a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small3_fwd<Small3Cfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small3_fwd<Small3Cfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small3_fwd<Small3Cfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}

// the stash of a forward that kept none, from its block outputs (lsnf_api.hip lsnf_restash); hipErrorInvalidValue = not covered
hipError_t lsnf_launch_small3_restash(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                      float* act_saved, int vec4, hipStream_t stream) {
    RestashArgs a;
    a.consts = plan + g.off_fwd_const; a.panels3b = plan + g.off_f3b_panels;
    a.z_out = z_out; a.z_saved = z_saved; a.act_saved = act_saved;
    a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    auto go = [&](auto cfg) -> hipError_t {
        using C = decltype(cfg);
        const size_t lds = ((size_t)(C::HT + 2 * C::WT) * S3_BTILE_FLOATS + 2 * C::WT * 32 + C::CONST_FLOATS) * sizeof(float);
        auto kern = lsnf_small3_restash_kernel<C>;
        static unsigned long long lds_ok = 0;
        if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)((B + S3_SAMPLES - 1) / S3_SAMPLES), (unsigned)g.depth), dim3(256), lds, stream, a);
        return hipGetLastError();
    };
    if (g.HT == 1 && g.WT == 1) return go(Small3Cfg<1, 1>{});
    if (g.HT == 2 && g.WT == 2) return go(Small3Cfg<2, 2>{});
    if (g.HT == 2 && g.WT == 4) return go(Small3Cfg<2, 4>{});
    return hipErrorInvalidValue;
}
