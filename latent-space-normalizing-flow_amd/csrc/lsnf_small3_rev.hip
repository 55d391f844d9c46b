// lsnf_small3_rev.hip -- latency reverse (sampling) pass on the bf16 matrix pipe: the L16 / bf16x3 scheme of
// lsnf_small3_fwd.hip (16-sample workgroups, producer-side split) for reference model.py:484-498 / :424-456.
// Per block, last to first; wave w owns the half-units z1[w], z2[w] of the running latent (registers):
//   R2,R3,R4 : h = f(z1) -> t[w], p[w]                              (forward panels S2..S4, 16x16x32 operand order)
//   CI       : z2 = z2 / sigmoid(p) - t ; objective -= sum log sigmoid(p)     (model.py:436-438), in registers
//   I1       : z = ([z1,z2] @ W^-1) * exp(-3 logs) - b ; objective -= log|det W| + sum 3 logs  (:193-196, 270, 246)
#include "lsnf_small3.h"

namespace {

template <int HT_, int WT_>
struct Small3RevCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = LSNF_FRAG3_FLOATS;
    static constexpr int OFF3_S2 = F * S::P1 * S::KT1;
    static constexpr int OFF3_S3 = OFF3_S2 + F * S::P2 * S::KT2;
    static constexpr int OFF3_S4 = OFF3_S3 + F * S::P3 * S::KT3;
    static constexpr int BLOCK3 = OFF3_S4 + F * S::P4 * S::KT4;
    static constexpr int BLOCKI = F * S::NZT * S::NZT;
    static constexpr int CONST_PER_BLOCK = S::FWD_CONST + S::INV_CONST;
    static constexpr int NU2 = (2 * WT_ + 3) / 4;
    // LDS map (floats): U = [z1 | z2 after the inverse coupling] (NZT B-tiles), H1, H2 (WT B-tiles each), reductions, constants
    static constexpr int L_U = 0;
    static constexpr int L_H1 = L_U + S::NZT * S3_BTILE_FLOATS;
    static constexpr int L_H2 = L_H1 + WT_ * S3_BTILE_FLOATS;
    static constexpr int L_RED = L_H2 + WT_ * S3_BTILE_FLOATS;
    static constexpr int L_CONST = L_RED + 4 * 16;
};

struct Small3RevArgs {
    const float* fwd_consts; const float* inv_consts; const float* panels3b; const float* ipanels3b;
    const float* z_in; const float* objective; float* z_out; float* objective_out;
    int B, nz, half, depth, vec4;
};

template <class C>
__global__ __launch_bounds__(256, 1) void lsnf_small3_rev_kernel(const Small3RevArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* U = smem + C::L_U;
    float* H1B = smem + C::L_H1;
    float* H2B = smem + C::L_H2;
    float* RED = smem + C::L_RED;
    float* cst = smem + C::L_CONST;
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
    const float* gf_last = a.panels3b + (size_t)last * C::BLOCK3;
    UFrags<HT> w2[NU2];
    UFrags<WT> w3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) {
        w2[i] = fetch_unit<HT>(gf_last + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
        w3[i] = fetch_unit<WT>(gf_last + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
    }
    const long sample = (long)blockIdx.x * S3_SAMPLES + n;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    // this wave's half-units of the input; z1 goes to LDS (R2's and I1's operand).  (Requested BEFORE the constant blocks are
    // copied: that copy waits for its loads in order, i.e. one memory round trip that the rows would otherwise start behind.)
    f32x4 z1 = load_row_half<HT>(nt1, ft1, a.z_in + row * (long)a.nz, a.half, g, vec4);
    f32x4 z2 = load_row_half<HT>(HT + nt1, ft1, a.z_in + row * (long)a.nz, a.half, g, vec4);
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < a.depth * C::CONST_PER_BLOCK; i += 256) {
        const int blk = i / C::CONST_PER_BLOCK, r = i % C::CONST_PER_BLOCK;
        cst[i] = r < C::FWD_CONST ? a.fwd_consts[blk * C::FWD_CONST + r] : a.inv_consts[blk * C::INV_CONST + (r - C::FWD_CONST)];
    }
    if (has1) store_half(U + nt1 * S3_BTILE_FLOATS, ft1, z1, lane);
    float obj = (wave == 0 && a.objective) ? a.objective[row] : 0.0f;      // per-wave partial
    __syncthreads();

    for (int blk = last; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_PER_BLOCK;
        const float* ci = cb + C::FWD_CONST;
        const float* gf = a.panels3b + (size_t)blk * C::BLOCK3;
        const float* gi = a.ipanels3b + (size_t)blk * C::BLOCKI;
        const int nb = blk > 0 ? blk - 1 : 0;                              // block 0 re-fetches its own panels: no loads under a branch
        const float* gfn = a.panels3b + (size_t)nb * C::BLOCK3;

        // ---- R2: h1 = relu(W1'^T z1 + c1) ----
        UFrags<WT> w4t = fetch_unit<WT>(gf + C::OFF3_S4, nt1, ft1, lane);
        UFrags<WT> w4p = fetch_unit<WT>(gf + C::OFF3_S4, HT + nt1, ft1, lane);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            const f32x4 h = relu4(unit_mma<HT>(unit_bias(cb + 32 * (C::P1 + nt), ft, g), w2[i], U, lane));
            if (hasw[i]) store_half(H1B + nt * S3_BTILE_FLOATS, ft, h, lane);
        }
        __syncthreads();
        // ---- R3: h2 ----
        UFrags<NZT> wia = fetch_unit<NZT>(gi, nt1, ft1, lane);
        UFrags<NZT> wib = fetch_unit<NZT>(gi, HT + nt1, ft1, lane);
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            const f32x4 h = relu4(unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + nt), ft, g), w3[i], H1B, lane));
            if (hasw[i]) store_half(H2B + nt * S3_BTILE_FLOATS, ft, h, lane);
        }
        __syncthreads();
        // ---- R4 + inverse coupling, in registers ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) w2[i] = fetch_unit<HT>(gfn + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
        const f32x4 tt_ = unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + nt1), ft1, g), w4t, H2B, lane);
        const f32x4 pp = unit_mma<WT>(unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + HT + nt1), ft1, g), w4p, H2B, lane);
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sig, lsig;
            lsnf_sigmoid_logsig(pp[r], sig, lsig);
            z2[r] = z2[r] / sig - tt_[r];
            lsum += lsig;
        }
        if (has1) {
            obj = obj - group_sum(lsum);
            store_half(U + (HT + nt1) * S3_BTILE_FLOATS, ft1, z2, lane);
        }
        if (wave == 0) {
            obj = obj - cb[32 * C::NP + 1];   // logdet - log|det W|       (model.py:196)
            obj = obj - cb[32 * C::NP + 0];   // logdet - sum(3 logs)      (model.py:273-276, reverse)
        }
        __syncthreads();
        // ---- I1: z = Winv'^T [z1; z2] + cinv ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) w3[i] = fetch_unit<WT>(gfn + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
        z1 = unit_mma<NZT>(unit_bias(ci + 32 * nt1, ft1, g), wia, U, lane);
        z2 = unit_mma<NZT>(unit_bias(ci + 32 * (HT + nt1), ft1, g), wib, U, lane);
        __syncthreads();                       // every wave has read U
        if (has1 && blk > 0) store_half(U + nt1 * S3_BTILE_FLOATS, ft1, z1, lane);
        __syncthreads();
    }
    if (has1 && live) {
        float* zr = a.z_out + sample * (long)a.nz;
        store_row_half<HT>(nt1, ft1, z1, zr, a.half, g, vec4);
        store_row_half<HT>(HT + nt1, ft1, z2, zr, a.half, g, vec4);
    }
    if (a.objective_out) {                     // kernel-uniform
        if (g == 0) RED[wave * 16 + n] = obj;
        __syncthreads();
        if (wave == 0 && g == 0 && live) a.objective_out[sample] = RED[n] + RED[16 + n] + RED[32 + n] + RED[48 + n];
    }
}

template <class C>
hipError_t launch_small3_rev(const Small3RevArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)C::L_CONST + (size_t)a.depth * C::CONST_PER_BLOCK) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_small3_rev_kernel<C>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + S3_SAMPLES - 1) / S3_SAMPLES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}
}  // namespace

hipError_t lsnf_launch_small3_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                      float* z_out, float* objective_out, int vec4, hipStream_t stream) {
    Small3RevArgs a;
    a.fwd_consts = plan + g.off_fwd_const; a.inv_consts = plan + g.off_inv_const;
    a.panels3b = plan + g.off_f3b_panels; a.ipanels3b = plan + g.off_i3b_panels;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.objective_out = objective_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_small3_rev<Small3RevCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_small3_rev<Small3RevCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_small3_rev<Small3RevCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
