// lsnf_rev3.hip -- throughput reverse (sampling) pass on the bf16 matrix pipe: the bf16x3 split and L16 lane layout of
// lsnf_fwd3.hip's 16x16x32 kernel applied to lsnf_rev.hip (replaces reference model.py:484-498 / :424-456).
// Per block, last to first, a wave owning 32 samples and all features:
//   R2,R3,R4 : h = f(z1) -> [t; p]                                (forward panels S2..S4, 16x16x32 operand order)
//   CI       : z2 = z2 / sigmoid(p) - t ; objective -= sum log sigmoid(p)     (model.py:436-438)
//   I1       : z = ([z1,z2] @ W^-1) * exp(-3 logs) - b ; objective -= log|det W| + sum 3 logs  (:193-196, 270, 246)
#include <stdlib.h>
#include "lsnf_l16.h"

namespace {

template <int HT_, int WT_>
struct Rev3Cfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    static constexpr int F = L16_FRAG_FLOATS;
    static constexpr int OFF3_S2 = F * S::P1 * S::KT1;
    static constexpr int OFF3_S3 = OFF3_S2 + F * S::P2 * S::KT2;
    static constexpr int OFF3_S4 = OFF3_S3 + F * S::P3 * S::KT3;
    static constexpr int BLOCK3 = OFF3_S4 + F * S::P4 * S::KT4;
    static constexpr int BLOCKI = F * S::NZT * S::NZT;
    static constexpr int CONST_PER_BLOCK = S::FWD_CONST + S::INV_CONST;
    static constexpr int SLOT3 = 2 * S::MAXKT * F;
};

struct Rev3Args {
    const float* fwd_consts; const float* inv_consts; const float* panels3b; const float* ipanels3b;
    const float* z_in; const float* objective; float* z_out; float* objective_out;
    int B, nz, half, depth, vec4;
    const unsigned* guard; int fixup;        // fp16 range guard, as in lsnf_fwd3.hip (Fwd3Args); the flag travels in z_out[first row of the wave][0]
};

template <class C, int NW>
__global__ __launch_bounds__(64 * NW, 1) void lsnf_rev3_kernel(const Rev3Args a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                         // depth * CONST_PER_BLOCK
    float* buf0 = smem + a.depth * C::CONST_PER_BLOCK;         // 2 x SLOT3
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int vec4 = a.vec4;
    const long wbase = ((long)blockIdx.x * NW + wave) * 32;           // first row of this wave
#if LSNF_L16_PARTS == 3
    if (a.fixup) {                                // fix-up pass of the fp16 reverse: only workgroups in which a wave raised its flag
        unsigned* fl = reinterpret_cast<unsigned*>(smem);
        unsigned f = 0u;
        if (wbase < a.B) f = __builtin_bit_cast(unsigned, a.z_out[wbase * (long)a.nz]) == LSNF_F16_SENTINEL_BITS ? 1u : 0u;
        if (lane == 0) fl[wave] = f;
        __syncthreads();
        unsigned any = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) any |= fl[w];
        __syncthreads();                          // (smem is reused below)
        if (any == 0u) return;                    // workgroup-uniform
    }
#else
    if (a.guard[0] != 0u) {                       // weights outside fp16's range (prepare): leave every row to the fix-up pass
        if (lane == 0 && wbase < a.B) a.z_out[wbase * (long)a.nz] = __builtin_bit_cast(float, LSNF_F16_SENTINEL_BITS);
        return;
    }
    bool bad = false;                             // an operand beyond fp16's range turns every output of its GEMM into NaN
#endif

    const int last = a.depth - 1;
    Pipe3<NW> pipe;
    pipe.buf0 = buf0; pipe.slot = C::SLOT3; pipe.wave = wave; pipe.lane = lane;
    pipe.template prime<first_kib(C::P2, C::KT2)>(a.panels3b + (size_t)last * C::BLOCK3 + C::OFF3_S2);

    const long base = ((long)blockIdx.x * NW + wave) * 32;
    long sample[2], rows[2]; bool live[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) { sample[st] = base + 16 * st + n; live[st] = sample[st] < a.B; rows[st] = live[st] ? sample[st] : (long)a.B - 1; }

    f32x16 x[NZT];
#pragma unroll
    for (int t = 0; t < NZT; ++t) x[t] = l16_load_tile<HT>(t, a.z_in, rows, a.nz, a.half, g, vec4);
    __builtin_amdgcn_sched_barrier(0);
    // (the constant blocks are copied AFTER the row loads have gone out: the copy waits for its loads in order, one memory
    //  round trip that the rows would otherwise start behind)
    for (int i = tid; i < a.depth * C::CONST_PER_BLOCK; i += 64 * NW) {
        const int blk = i / C::CONST_PER_BLOCK, r = i % C::CONST_PER_BLOCK;
        cst[i] = r < C::FWD_CONST ? a.fwd_consts[blk * C::FWD_CONST + r] : a.inv_consts[blk * C::INV_CONST + (r - C::FWD_CONST)];
    }
    float obj[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) obj[st] = a.objective ? a.objective[rows[st]] : 0.0f;
#if LSNF_L16_PARTS == 3
    auto keep = [](f32x16 acc, int) { return acc; };
    auto relu = [](f32x16 acc, int) { return lsnf_relu16(acc); };
#else
    auto keep = [&](f32x16 acc, int t) { if (t == 0) bad = bad || acc[0] != acc[0] || acc[4] != acc[4]; return acc; };
    auto relu = [&](f32x16 acc, int t) { if (t == 0) bad = bad || acc[0] != acc[0] || acc[4] != acc[4]; return lsnf_relu16(acc); };
#endif

    for (int blk = last; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_PER_BLOCK;
        const float* ci = cb + C::FWD_CONST;
        const float* gf = a.panels3b + (size_t)blk * C::BLOCK3;
        const float* gi = a.ipanels3b + (size_t)blk * C::BLOCKI;
        const float* gnext = blk > 0 ? a.panels3b + (size_t)(blk - 1) * C::BLOCK3 + C::OFF3_S2 : nullptr;

        // ---- R2: h1 = relu(W1'^T z1 + c1) ----
        f32x16 h1[WT];
        {
            Split3 zs[2 * HT];
            l16_split_tiles<HT>(x, zs);
            l16_gemm_stage3<C::P2, C::KT2, first_kib(C::P3, C::KT3)>(
                pipe, gf + C::OFF3_S2, gf + C::OFF3_S3, h1, zs, [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + t), g); }, relu);
        }
        // ---- R3 ----
        f32x16 h2[WT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(h1, hs);
            l16_gemm_stage3<C::P3, C::KT3, first_kib(C::P4, C::KT4)>(
                pipe, gf + C::OFF3_S3, gf + C::OFF3_S4, h2, hs,
                [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + C::P2 + t), g); }, relu);
        }
        // ---- R4: [t; p] ----
        f32x16 tp[2 * HT];
        {
            Split3 hs[2 * WT];
            l16_split_tiles<WT>(h2, hs);
            l16_gemm_stage3<C::P4, C::KT4, first_kib(NZT, NZT)>(
                pipe, gf + C::OFF3_S4, gi, tp, hs,
                [&](int t) { return l16_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + t), g); }, keep);
        }
        // ---- inverse coupling (model.py:436-438) ----
        float lsum[2] = {0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(tp[HT + t][r], sig, lsig);
                x[HT + t][r] = x[HT + t][r] / sig - tp[t][r];
                lsum[(r >> 2) & 1] += lsig;
            }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            obj[st] = obj[st] - l16_group_sum(lsum[st]);
            obj[st] = obj[st] - cb[32 * C::NP + 1];   // logdet - log|det W|       (model.py:196)
            obj[st] = obj[st] - cb[32 * C::NP + 0];   // logdet - sum(3 logs)      (model.py:273-276, reverse)
        }
        // ---- I1: z = Winv'^T [z1; z2] + cinv ----
        {
            Split3 us[2 * NZT];
            l16_split_tiles<NZT>(x, us);
            f32x16 xn[NZT];
            l16_gemm_stage3<NZT, NZT, first_kib(C::P2, C::KT2)>(
                pipe, gi, gnext, xn, us, [&](int t) { return l16_bias_init(ci + 32 * t, g); }, keep);
#pragma unroll
            for (int t = 0; t < NZT; ++t) x[t] = xn[t];
        }
    }
#if LSNF_L16_PARTS == 2
    // this wave met an operand beyond fp16's range: element [first row][0] of its output (register 0 of tile 0 on lane 0)
    // carries the flag for the bf16x3 fix-up pass queued behind this kernel
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) x[0][0] = __builtin_bit_cast(float, LSNF_F16_SENTINEL_BITS);
#endif
#pragma unroll
    for (int t = 0; t < NZT; ++t) l16_store_tile<HT>(t, x[t], a.z_out, sample, live, a.nz, a.half, g, vec4);
    if (a.objective_out) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
            if (live[st] && g == 0) a.objective_out[sample[st]] = obj[st];
    }
}

template <class C, int NW>
hipError_t launch_rev3_w(const Rev3Args& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.depth * C::CONST_PER_BLOCK + 2 * (size_t)C::SLOT3) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_rev3_kernel<C, NW>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * NW - 1) / (32 * NW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}
template <class C>
hipError_t launch_rev3(const Rev3Args& a, hipStream_t stream) {
    static const char* fw = getenv("LSNF_FORCE_WAVES");   // experiment knob (tools/): 4 or 8
    const bool eight = fw ? atoi(fw) == 8 : a.B > 128 * 256;
    return eight ? launch_rev3_w<C, 8>(a, stream) : launch_rev3_w<C, 4>(a, stream);
}
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); hipErrorInvalidValue = not covered (e.g. very deep stacks: LDS)
#ifndef LSNF_REV3_ENTRY
#define LSNF_REV3_ENTRY lsnf_launch_reverse3
#endif
hipError_t LSNF_REV3_ENTRY(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                           float* z_out, float* objective_out, int vec4, int fixup, hipStream_t stream) {
    Rev3Args a;
    a.guard = reinterpret_cast<const unsigned*>(plan + g.off_guard); a.fixup = fixup;
    a.fwd_consts = plan + g.off_fwd_const; a.inv_consts = plan + g.off_inv_const;
#if LSNF_L16_PARTS == 3
    a.panels3b = plan + g.off_f3b_panels; a.ipanels3b = plan + g.off_i3b_panels;
#else
    a.panels3b = plan + g.off_f2h_panels; a.ipanels3b = plan + g.off_i2h_panels;
#endif
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.objective_out = objective_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_rev3<Rev3Cfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_rev3<Rev3Cfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_rev3<Rev3Cfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
