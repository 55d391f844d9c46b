// lsnf_small3_rev.hip -- latency reverse (sampling) pass on the bf16 matrix pipe: the L16 / bf16x3 scheme of
// lsnf_small3_fwd.hip (workgroups of ST sample tiles of 16 rows, producer-side split, weights re-loaded in place one block ahead:
// lsnf_small3.h units_mma_st) for reference model.py:484-498 / :424-456.
// Per block, last to first; wave w owns the half-units z1[w], z2[w] of the running latent (registers):
//   R2,R3,R4 : h = f(z1) -> t[w], p[w]                              (forward panels S2..S4, 16x16x32 operand order)
//   CI       : z2 = z2 / sigmoid(p) - t ; objective -= sum log sigmoid(p)     (model.py:436-438), in registers: R4's epilogue
//   I1       : z = ([z1,z2] @ W^-1) * exp(-3 logs) - b ; objective -= log|det W| + sum 3 logs  (:193-196, 270, 246);
//              its epilogue publishes the new z1 (the next block's R2 operand) into the OTHER z1 buffer: four barriers per block
#include <stdlib.h>
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
};
// LDS map (floats) for ST sample tiles: z1 tiles (HT per sample tile, double-buffered: I1 publishes the next block's z1 while it
// reads this block's), z2 tiles after the inverse coupling (HT), H1, H2 (WT each), reductions, constants
template <class C, int ST>
struct Small3RevLds {
    static constexpr int GH = C::HT * S3_BTILE_FLOATS, HL = C::WT * S3_BTILE_FLOATS;
    static constexpr int L_Z1 = 0;
    static constexpr int L_Z2 = L_Z1 + 2 * ST * GH;
    static constexpr int L_H1 = L_Z2 + ST * GH;
    static constexpr int L_H2 = L_H1 + ST * HL;
    static constexpr int L_RED = L_H2 + ST * HL;
    static constexpr int L_CONST = L_RED + ST * 4 * 16;
};

struct Small3RevArgs {
    const float* fwd_consts; const float* inv_consts; const float* panels3b; const float* ipanels3b;
    const float* z_in; const float* objective; float* z_out; float* objective_out;
    int B, nz, half, depth, vec4;
};

template <class C, int ST>
__global__ __launch_bounds__(256, 1) void lsnf_small3_rev_kernel(const Small3RevArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2, LASTU = NU2 - 1;
    using L = Small3RevLds<C, ST>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Z1 = smem + L::L_Z1;
    float* Z2 = smem + L::L_Z2;
    float* H1B = smem + L::L_H1;
    float* H2B = smem + L::L_H2;
    float* RED = smem + L::L_RED;
    float* cst = smem + L::L_CONST;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int vec4 = a.vec4;

    // (a wave without a unit of its own computes unit 0 again and stores the same values to the same LDS words: no branch in the stages)
    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }

    const int last = a.depth - 1;
    const float* gf_last = a.panels3b + (size_t)last * C::BLOCK3;
    const float* gi_last = a.ipanels3b + (size_t)last * C::BLOCKI;
    long sample[ST]; bool live[ST]; long row[ST];
    f32x4 z1[ST], z2[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        sample[st] = ((long)blockIdx.x * ST + st) * S3_SAMPLES + n;
        live[st] = sample[st] < a.B;
        row[st] = live[st] ? sample[st] : (long)a.B - 1;
        // this wave's half-units of the input (requested before the weights and the constants: vmcnt completes in order)
        z1[st] = load_row_half<HT>(nt1, ft1, a.z_in + row[st] * (long)a.nz, a.half, g, vec4);
        z2[st] = load_row_half<HT>(HT + nt1, ft1, a.z_in + row[st] * (long)a.nz, a.half, g, vec4);
    }
    float obj[ST];                                                        // per-wave partial of the running objective
#pragma unroll
    for (int st = 0; st < ST; ++st) obj[st] = (wave == 0 && a.objective) ? a.objective[row[st]] : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    UFrags<HT> w2[NU2];
    UFrags<WT> w3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) w2[i] = fetch_unit<HT>(gf_last + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
#pragma unroll
    for (int i = 0; i < NU2; ++i) w3[i] = fetch_unit<WT>(gf_last + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
    UFrags<WT> w4t = fetch_unit<WT>(gf_last + C::OFF3_S4, nt1, ft1, lane);
    UFrags<WT> w4p = fetch_unit<WT>(gf_last + C::OFF3_S4, HT + nt1, ft1, lane);
    UFrags<NZT> wia = fetch_unit<NZT>(gi_last, nt1, ft1, lane);
    UFrags<NZT> wib = fetch_unit<NZT>(gi_last, HT + nt1, ft1, lane);
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < a.depth * C::CONST_PER_BLOCK; i += 256) {
        const int blk = i / C::CONST_PER_BLOCK, r = i % C::CONST_PER_BLOCK;
        cst[i] = r < C::FWD_CONST ? a.fwd_consts[blk * C::FWD_CONST + r] : a.inv_consts[blk * C::INV_CONST + (r - C::FWD_CONST)];
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) store_half(Z1 + st * L::GH + nt1 * S3_BTILE_FLOATS, ft1, z1[st], lane);   // buffer 0
    float lsum[ST];                                                       // running sum of log sigmoid(p) over all blocks
#pragma unroll
    for (int st = 0; st < ST; ++st) lsum[st] = 0.0f;
    __syncthreads();

    for (int blk = last; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_PER_BLOCK;
        const float* ci = cb + C::FWD_CONST;
        const float* gi = a.ipanels3b + (size_t)blk * C::BLOCKI;
        const int nb = blk > 0 ? blk - 1 : 0;                              // block 0 re-fetches its own panels: no loads under a branch
        const float* gfn = a.panels3b + (size_t)nb * C::BLOCK3;
        const float* gin = a.ipanels3b + (size_t)nb * C::BLOCKI;
        float* z1_cur = Z1 + ((last - blk) & 1) * ST * L::GH;
        float* z1_nxt = Z1 + ((last - blk + 1) & 1) * ST * L::GH;

        // ---- R2: h1 = relu(W1'^T z1 + c1) ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 h[ST];
            const f32x4 bh = unit_bias(cb + 32 * (C::P1 + nt), ft, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) h[st] = bh;
            auto epi = [&](int st) { h[st] = relu4(h[st]); store_half(H1B + st * L::HL + nt * S3_BTILE_FLOATS, ft, h[st], lane); };
            const bf16x8* rf = unit_ptr<HT>(gfn + C::OFF3_S2, nt, ft, lane);
            if (i == 0) {        // carry: the last k-tile of I1's two units, THIS block's fragments (I1 runs last in the block)
                const bf16x8* ca = unit_ptr<NZT>(gi, nt1, ft1, lane);
                const bf16x8* cbp = unit_ptr<NZT>(gi, HT + nt1, ft1, lane);
                units_mma_st<HT, 0, HT, ST, 1, 28, 6, 0>(h, h, w2[i], w2[i], rf, nullptr, z1_cur, L::GH, lane, epi,
                    [&](int q) { if (q < 3) refill_last<NZT>(wia, ca, q); else refill_last<NZT>(wib, cbp, q - 3); }, [](int) {});
            } else {
                const bf16x8* cp = unit_ptr<HT>(gfn + C::OFF3_S2, hw[i > 0 ? i - 1 : 0] >> 1, hw[i > 0 ? i - 1 : 0] & 1, lane);
                units_mma_st<HT, 0, HT, ST, 1, 28, 3, 0>(h, h, w2[i], w2[i], rf, nullptr, z1_cur, L::GH, lane, epi,
                    [&](int q) { refill_last<HT>(w2[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
        }
        __syncthreads();
        // ---- R3: h2 ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 h[ST];
            const f32x4 bh = unit_bias(cb + 32 * (C::P1 + C::P2 + nt), ft, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) h[st] = bh;
            auto epi = [&](int st) { h[st] = relu4(h[st]); store_half(H2B + st * L::HL + nt * S3_BTILE_FLOATS, ft, h[st], lane); };
            const bf16x8* rf = unit_ptr<WT>(gfn + C::OFF3_S3, nt, ft, lane);
            if (i == 0) {
                const bf16x8* cp = unit_ptr<HT>(gfn + C::OFF3_S2, hw[LASTU] >> 1, hw[LASTU] & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 28, 3, 0>(h, h, w3[i], w3[i], rf, nullptr, H1B, L::HL, lane, epi,
                    [&](int q) { refill_last<HT>(w2[LASTU], cp, q); }, [](int) {});
            } else {
                const bf16x8* cp = unit_ptr<WT>(gfn + C::OFF3_S3, hw[i > 0 ? i - 1 : 0] >> 1, hw[i > 0 ? i - 1 : 0] & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 28, 3, 0>(h, h, w3[i], w3[i], rf, nullptr, H1B, L::HL, lane, epi,
                    [&](int q) { refill_last<WT>(w3[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
        }
        __syncthreads();
        // ---- R4 + inverse coupling (its epilogue): z2 = z2 / sigmoid(p) - t ----
        {
            f32x4 tt_[ST], pp[ST];
            const f32x4 bt = unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + nt1), ft1, g);
            const f32x4 bp = unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + HT + nt1), ft1, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) { tt_[st] = bt; pp[st] = bp; }
            const bf16x8* c3 = unit_ptr<WT>(gfn + C::OFF3_S3, hw[LASTU] >> 1, hw[LASTU] & 1, lane);
            units_mma_st<WT, 0, WT, ST, 2, 60, 3, 0>(tt_, pp, w4t, w4p, unit_ptr<WT>(gfn + C::OFF3_S4, nt1, ft1, lane),
                unit_ptr<WT>(gfn + C::OFF3_S4, HT + nt1, ft1, lane), H2B, L::HL, lane,
                [&](int st) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sig, lsig;
                        lsnf_sigmoid_logsig(pp[st][r], sig, lsig);
                        z2[st][r] = z2[st][r] / sig - tt_[st][r];
                        lsum[st] += lsig;
                    }
                    store_half(Z2 + st * L::GH + nt1 * S3_BTILE_FLOATS, ft1, z2[st], lane);
                },
                [&](int q) { refill_last<WT>(w3[LASTU], c3, q); }, [](int) {});
        }
        if (wave == 0) {
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                obj[st] = obj[st] - cb[32 * C::NP + 1];   // logdet - log|det W|       (model.py:196)
                obj[st] = obj[st] - cb[32 * C::NP + 0];   // logdet - sum(3 logs)      (model.py:273-276, reverse)
            }
        }
        __syncthreads();
        // ---- I1: z = Winv'^T [z1; z2] + cinv; the new z1 goes to the other z1 buffer under the last steps ----
        {
            const f32x4 b1 = unit_bias(ci + 32 * nt1, ft1, g), b2 = unit_bias(ci + 32 * (HT + nt1), ft1, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) { z1[st] = b1; z2[st] = b2; }
            const bf16x8* c4t = unit_ptr<WT>(gfn + C::OFF3_S4, nt1, ft1, lane);
            const bf16x8* c4p = unit_ptr<WT>(gfn + C::OFF3_S4, HT + nt1, ft1, lane);
            units_mma_st<NZT, 0, NZT, ST, 2, 24, 6, 0, HT>(z1, z2, wia, wib, unit_ptr<NZT>(gin, nt1, ft1, lane), unit_ptr<NZT>(gin, HT + nt1, ft1, lane),
                z1_cur, L::GH, lane,
                [&](int st) { store_half(z1_nxt + st * L::GH + nt1 * S3_BTILE_FLOATS, ft1, z1[st], lane); },   // (after block 0: nobody reads it)
                [&](int q) { if (q < 3) refill_last<WT>(w4t, c4t, q); else refill_last<WT>(w4p, c4p, q - 3); }, [](int) {}, Z2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        if (has1 && live[st]) {
            float* zr = a.z_out + sample[st] * (long)a.nz;
            store_row_half<HT>(nt1, ft1, z1[st], zr, a.half, g, vec4);
            store_row_half<HT>(HT + nt1, ft1, z2[st], zr, a.half, g, vec4);
        }
    }
    if (a.objective_out) {                     // kernel-uniform
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const float o = obj[st] - (has1 ? group_sum(lsum[st]) : 0.0f);
            if (g == 0) RED[(st * 4 + wave) * 16 + n] = o;
        }
        __syncthreads();
        if (wave == 0 && g == 0) {
#pragma unroll
            for (int st = 0; st < ST; ++st)
                if (live[st]) a.objective_out[sample[st]] = RED[(st * 4 + 0) * 16 + n] + RED[(st * 4 + 1) * 16 + n] + RED[(st * 4 + 2) * 16 + n] + RED[(st * 4 + 3) * 16 + n];
        }
    }
}

template <class C, int ST>
hipError_t launch_small3_rev_st(const Small3RevArgs& a, hipStream_t stream) {
    if constexpr ((size_t)(Small3RevLds<C, ST>::L_CONST + C::CONST_PER_BLOCK) * sizeof(float) > 160 * 1024 || (ST == 4 && C::WT > 2)) {
        return hipErrorInvalidValue;
    } else {
        const size_t lds = ((size_t)Small3RevLds<C, ST>::L_CONST + (size_t)a.depth * C::CONST_PER_BLOCK) * sizeof(float);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        auto kern = lsnf_small3_rev_kernel<C, ST>;
        static unsigned long long lds_ok = 0;
        if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
        const unsigned grid = (unsigned)((a.B + ST * S3_SAMPLES - 1) / (ST * S3_SAMPLES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
        return hipGetLastError();
    }
}
// rows per workgroup by batch size, as the forward (lsnf_small3_fwd.hip launch_small3_fwd); LSNF_SMALL3_ST forces a shape
template <class C>
hipError_t launch_small3_rev(const Small3RevArgs& a, hipStream_t stream) {
    static const char* env = getenv("LSNF_SMALL3_ST");
    const int st = env ? atoi(env) : (a.B <= 256 * 16 ? 1 : (a.B <= 256 * 32 ? 2 : 4));
    hipError_t e = hipErrorInvalidValue;
    if (st >= 4) e = launch_small3_rev_st<C, 4>(a, stream);
    if (e == hipErrorInvalidValue && st >= 2) e = launch_small3_rev_st<C, 2>(a, stream);
    if (e == hipErrorInvalidValue) e = launch_small3_rev_st<C, 1>(a, stream);
    return e;
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
