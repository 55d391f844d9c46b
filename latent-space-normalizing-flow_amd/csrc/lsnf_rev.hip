// lsnf_rev.hip -- fused reverse (sampling) pass of the whole stack in one launch.
// Replaces reference model.py:484-498 (_netF.forward(reverse=True)) -> :361-363 -> :424-456:
// per block, last to first:  h = f(z1); z2 = z2/scale - shift; logdet -= sum log scale;
// z = [z1,z2] @ W^-1; logdet -= log|det W|; actnorm^-1 (x*exp(-logs) - b); logdet -= sum logs.
// Same sample-on-lane MFMA scheme as lsnf_fwd.hip; the MLP panels are the forward stream's,
// the W^-1 panels (with exp(-3 logs) and -b folded) come from the inverse stream.
#include "lsnf_device.h"

namespace {

template <int HT_, int WT_>
struct RevCfg : LsnfStackCfg<HT_, WT_> {
    using S = LsnfStackCfg<HT_, WT_>;
    using S::HT; using S::WT; using S::NZT;
    static constexpr int CONST_PER_BLOCK = S::FWD_CONST + S::INV_CONST;
};

struct RevArgs {
    const float* fwd_consts; const float* fwd_panels; const float* inv_consts; const float* inv_panels;
    const float* z_in; const float* objective; float* z_out; float* objective_out;
    int B, nz, half, depth, vec4;
};

// NW waves per workgroup (4: two workgroups per CU, 8: one; see lsnf_fwd.hip)
template <class C, int NW>
__global__ __launch_bounds__(64 * NW, 2) void lsnf_rev_kernel(const RevArgs a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                   // depth * (FWD_CONST + INV_CONST)
    const int tid = threadIdx.x;
    LsnfPipeT<NW> pipe;
    pipe.buf0 = smem + a.depth * C::CONST_PER_BLOCK;
    pipe.slot = C::SLOT;
    pipe.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pipe.lane = tid & 63;
    const int lane = pipe.lane, m = lane & 31, h = lane >> 5;

    pipe.template prime<HT>(a.fwd_panels + (size_t)(a.depth - 1) * C::FWD_BLOCK + C::OFF_S2);
    for (int i = tid; i < a.depth * C::CONST_PER_BLOCK; i += 64 * NW) {
        const int blk = i / C::CONST_PER_BLOCK, r = i % C::CONST_PER_BLOCK;
        cst[i] = r < C::FWD_CONST ? a.fwd_consts[blk * C::FWD_CONST + r] : a.inv_consts[blk * C::INV_CONST + (r - C::FWD_CONST)];
    }
    const long sample = ((long)blockIdx.x * NW + pipe.wave) * 32 + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;
    f32x16 x[NZT];
    lsnf_load_rows<HT>(x, a.z_in, row, a.nz, a.half, h, a.vec4);
    float obj = a.objective ? a.objective[row] : 0.0f;

    for (int blk = a.depth - 1; blk >= 0; --blk) {
        const float* cb = cst + blk * C::CONST_PER_BLOCK;
        const float* ci = cb + C::FWD_CONST;
        const float* gf = a.fwd_panels + (size_t)blk * C::FWD_BLOCK;
        const float* gi = a.inv_panels + (size_t)blk * C::INV_BLOCK;
        const float* gnext = blk > 0 ? a.fwd_panels + (size_t)(blk - 1) * C::FWD_BLOCK + C::OFF_S2 : nullptr;

        // h1 = relu(W1'^T z1 + c1)
        f32x16 h1[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* nxt = (nt + 1 < WT) ? gf + C::OFF_S2 + (nt + 1) * HT * LSNF_FRAG_FLOATS : gf + C::OFF_S3;
            const float* lb = (nt + 1 < WT) ? pipe.template acquire<HT>(nxt) : pipe.template acquire<WT>(nxt);
            h1[nt] = lsnf_bias_init(cb + 32 * (C::P1 + nt), h);
            lsnf_panel_mma<HT>(h1[nt], x, lb, lane);
            h1[nt] = lsnf_relu16(h1[nt]);
        });
        f32x16 h2[WT];
        lsnf_static_for<WT>([&](auto nt) {
            const float* nxt = (nt + 1 < WT) ? gf + C::OFF_S3 + (nt + 1) * WT * LSNF_FRAG_FLOATS : gf + C::OFF_S4;
            const float* lb = pipe.template acquire<WT>(nxt);
            h2[nt] = lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + nt), h);
            lsnf_panel_mma<WT>(h2[nt], h1, lb, lane);
            h2[nt] = lsnf_relu16(h2[nt]);
        });
        f32x16 tp[2 * HT];
        lsnf_static_for<2 * HT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < 2 * HT) lb = pipe.template acquire<WT>(gf + C::OFF_S4 + (nt + 1) * WT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<NZT>(gi);
            tp[nt] = lsnf_bias_init(cb + 32 * (C::P1 + C::P2 + C::P3 + nt), h);
            lsnf_panel_mma<WT>(tp[nt], h2, lb, lane);
        });
        // z2 = z2 / scale - shift ; logdet -= sum log(scale)   (model.py:436-438)
        float lsum = 0.0f;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sig, lsig;
                lsnf_sigmoid_logsig(tp[HT + t][r], sig, lsig);
                x[HT + t][r] = x[HT + t][r] / sig - tp[t][r];
                lsum += lsig;
            }
        obj = obj - lsnf_pair_sum(lsum);
        // z = ([z1,z2] @ W^-1) * exp(-3 logs) - b   (model.py:193-194, 270, 246)
        f32x16 y[NZT];
        lsnf_static_for<NZT>([&](auto nt) {
            const float* lb;
            if constexpr (nt + 1 < NZT) lb = pipe.template acquire<NZT>(gi + (nt + 1) * NZT * LSNF_FRAG_FLOATS);
            else lb = pipe.template acquire<HT>(gnext);
            y[nt] = lsnf_bias_init(ci + 32 * nt, h);
            lsnf_panel_mma<NZT>(y[nt], x, lb, lane);
        });
#pragma unroll
        for (int t = 0; t < NZT; ++t) x[t] = y[t];
        obj = obj - cb[32 * C::NP + 1];   // logdet - dlogdet          (model.py:196)
        obj = obj - cb[32 * C::NP + 0];   // logdet + (-1)*sum(3 logs)  (model.py:273-276)
    }
    if (live) {
        lsnf_store_rows<HT>(x, a.z_out, sample, a.nz, a.half, h, a.vec4);
        if (h == 0 && a.objective_out) a.objective_out[sample] = obj;
    }
}

template <class C, int NW>
hipError_t launch_rev_w(const RevArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)a.depth * C::CONST_PER_BLOCK + 2 * (size_t)C::SLOT) * sizeof(float);
    auto kern = lsnf_rev_kernel<C, NW>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * NW - 1) / (32 * NW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}
template <class C>
hipError_t launch_rev(const RevArgs& a, hipStream_t stream) {
    return a.B > 256 * 128 ? launch_rev_w<C, 8>(a, stream) : launch_rev_w<C, 4>(a, stream);
}
}  // namespace

hipError_t lsnf_launch_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                               float* z_out, float* objective_out, int vec4, hipStream_t stream) {
    RevArgs a;
    a.fwd_consts = plan + g.off_fwd_const; a.fwd_panels = plan + g.off_fwd_panels;
    a.inv_consts = plan + g.off_inv_const; a.inv_panels = plan + g.off_inv_panels;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.objective_out = objective_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.depth = g.depth; a.vec4 = vec4;
    if (g.HT == 1 && g.WT == 1) return launch_rev<RevCfg<1, 1>>(a, stream);
    if (g.HT == 2 && g.WT == 2) return launch_rev<RevCfg<2, 2>>(a, stream);
    if (g.HT == 2 && g.WT == 4) return launch_rev<RevCfg<2, 4>>(a, stream);
    return hipErrorInvalidValue;
}
