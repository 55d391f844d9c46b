// lsnf_params.hip -- parameter gradients of the flow stack (replaces autograd of train.py:406-411,
// `loss_f.backward()` into the 60 live tensors of _netF).
//
// Three steps on one stream:
//  (1) lsnf_bwd_z_kernel<DUMP>  (lsnf_bwd.hip): the fused backward, which additionally writes, per block,
//      the per-sample gradients at every GEMM output (g_v, g_a1, g_a2, g_t, g_p) and the recomputed
//      hidden activations (h1, h2), and accumulates G = sum_b dL/dlogdet_b.
//  (2) lsnf_tn_gemm_kernel: dM = A^T G contracted over the batch (fp32 MFMA, operands loaded straight
//      from HBM in fragment layout -- rows are contiguous along the feature index, which is the lane
//      index of both operands), plus the column sums of G (bias gradients); split over sample chunks,
//      accumulated with float atomics into the zero-initialised "folded" gradient buffer.
//  (3) lsnf_unfold_kernel: chain rule from the folded matrices (Wa = diag(e^{3s}) W, W1' = W1 diag(e^{3s1}),
//      ..., interleaved fc_zeros columns) back to the reference's raw parameters, incl.
//      d log|det W| / dW = W^-T (model.py:182) and d sum(3 logs)/d logs = 3 (model.py:264,273).
#include <hip/hip_runtime.h>
#include "lsnf_layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LsnfParamPtrs { const float* p[LSNF_MAX_DEPTH * 12]; };
struct LsnfGradPtrs { float* p[LSNF_MAX_DEPTH * 12]; };
enum { P_AB = 0, P_ALOGS, P_W, P_W1, P_B1, P_LOGS1, P_W2, P_B2, P_LOGS2, P_W3, P_B3, P_LOGS3 };

hipError_t lsnf_launch_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                  const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                  float* g_z_in, float* dump, float* gl_total, int vec4, hipStream_t stream,
                                  const LsnfLangevinArgs* lv = nullptr, const float* act_saved = nullptr);
hipError_t lsnf_launch_small_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                        const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale, float* g_z_in,
                                        int vec4, hipStream_t stream, const LsnfLangevinArgs* lv, const float* act_saved,
                                        float* dump, float* gl_total);

namespace {

struct TnArgs {
    const float* z_in; const float* z_out; const float* z_saved;
    const float* dump; float* fold;
    int B, nz, half, width, depth, chunk;
};

// task = blk*5 + which: 0 dWa (A = block input x, G = g_v) ; 1 dW1' (A = v1, G = g_a1) ; 2 dW2' (A = h1, G = g_a2)
//                       3 dW3s (A = h2, G = g_t) ; 4 dW3p (A = h2, G = g_p)
__global__ __launch_bounds__(256) void lsnf_tn_gemm_kernel(const TnArgs a) {
    const int task = blockIdx.x, blk = task / 5, which = task % 5;
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const LsnfFoldLayout fl = lsnf_fold_layout(a.nz, a.width);
    const float* dmp = a.dump + (size_t)blk * dl.per_block;
    float* fold = a.fold + (size_t)blk * fl.per_block;
    const float* yblk = (blk == a.depth - 1) ? a.z_out : a.z_saved + (size_t)blk * a.B * a.nz;       // block output
    const float* xblk = (blk == 0) ? a.z_in : a.z_saved + (size_t)(blk - 1) * a.B * a.nz;            // block input
    const float *A, *G; int lda, ldg, K, N; float *C, *cs;
    switch (which) {
    case 0: A = xblk; lda = a.nz; K = a.nz; G = dmp + dl.off_gv; ldg = a.nz; N = a.nz; C = fold + fl.dWa; cs = fold + fl.dca; break;
    case 1: A = yblk; lda = a.nz; K = a.half; G = dmp + dl.off_ga1; ldg = a.width; N = a.width; C = fold + fl.dW1; cs = fold + fl.dc1; break;
    case 2: A = dmp + dl.off_h1; lda = a.width; K = a.width; G = dmp + dl.off_ga2; ldg = a.width; N = a.width; C = fold + fl.dW2; cs = fold + fl.dc2; break;
    case 3: A = dmp + dl.off_h2; lda = a.width; K = a.width; G = dmp + dl.off_gt; ldg = a.half; N = a.half; C = fold + fl.dW3s; cs = fold + fl.dc3s; break;
    default: A = dmp + dl.off_h2; lda = a.width; K = a.width; G = dmp + dl.off_gp; ldg = a.half; N = a.half; C = fold + fl.dW3p; cs = fold + fl.dc3p; break;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, kk = lane >> 5;
    const int kcol = 32 * wave + i;              // this wave owns k-tile `wave` (K <= 128 -> <= 4 tiles)
    if (32 * wave >= K) return;                  // wave-uniform
    const bool kok = kcol < K;
    const int NT = (N + 31) / 32;
    f32x16 acc[4];
    float csum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { csum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; }
    const int m_begin = blockIdx.y * a.chunk;
    const int m_end = min(a.B, m_begin + a.chunk);
    // U sample pairs per trip: all loads of a trip are issued before its MFMAs, so a trip costs one memory latency
    // (the per-pair dependent load -> MFMA chain made small batches latency-bound: 50 trips at B = 100)
    constexpr int U = 8;
    int ncol[4]; bool nok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ncol[t] = 32 * t + i; nok[t] = t < NT && ncol[t] < N; }
    for (int m0 = m_begin; m0 < m_end; m0 += 2 * U) {
        float av[U], gv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 2 * u + kk;
            const bool mok = m < m_end;
            av[u] = (mok && kok) ? A[(size_t)m * lda + kcol] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) gv[u][t] = (mok && nok[t]) ? G[(size_t)m * ldg + ncol[t]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < NT) {
                    csum[t] += gv[u][t];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], gv[u][t], acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < NT) {
            const int ncol = 32 * t + i;
            if (ncol < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (krow < K) atomicAdd(&C[(size_t)krow * N + ncol], acc[t][r]);
                }
            }
            if (wave == 0) {   // column sums of G (bias gradients), once per task
                const float s = csum[t] + __shfl_xor(csum[t], 32, 64);
                if (kk == 0 && ncol < N) atomicAdd(&cs[ncol], s);
            }
        }
    }
}

// raw-parameter gradients from the folded ones.  grid = (depth, UNFOLD_SECTIONS): the sections of one block share
// the element-wise loops (grid-strided), split the row reductions wave-per-row, and sections 0..2 take one
// column-reduction each (fc_1, fc_2, fc_zeros).
#define LSNF_UNFOLD_SECTIONS 8
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void lsnf_unfold_kernel(LsnfParamPtrs pp, LsnfGradPtrs gp, const float* fold_all,
                                                          const float* gl_total, const float* winv_all, int nz, int width,
                                                          int coupling) {
    const int blk = blockIdx.x, sec = blockIdx.y, tid = threadIdx.x;
    const int gtid = sec * 256 + tid, gstride = LSNF_UNFOLD_SECTIONS * 256;
    const int half = nz / 2, w = width;
    const LsnfFoldLayout fl = lsnf_fold_layout(nz, width);
    const float* F = fold_all + (size_t)blk * fl.per_block;
    const float* const* P = &pp.p[blk * 12];
    float* const* Gp = &gp.p[blk * 12];
    const float* winv = winv_all + (size_t)blk * nz * nz;
    const double Gtot = (double)gl_total[0];
    // exp(3*logs) of the four log-scale vectors, once (float64)
    __shared__ double ea[128], e1v[128], e2v[128], e3v[128];
    const int n3 = coupling ? nz : half;           // fc_zeros outputs: interleaved shift/scale columns, or shift only
    for (int k = tid; k < nz; k += 256) { ea[k] = exp((double)(P[P_ALOGS][k] * 3.0f)); if (k < n3) e3v[k] = exp((double)(P[P_LOGS3][k] * 3.0f)); }
    for (int k = tid; k < w; k += 256) { e1v[k] = exp((double)(P[P_LOGS1][k] * 3.0f)); e2v[k] = exp((double)(P[P_LOGS2][k] * 3.0f)); }
    __syncthreads();
    // ---- actnorm + 1x1 conv:  Wa = diag(e) W, ca = (b*e) W, const = 3*sum(s) + log|det W|
    // row reductions t1_k = sum_n W_kn dWa_kn, t2_k = sum_n W_kn dca_n: one wave per row k, lanes along n (coalesced)
    {
        const int lane = tid & 63, gwave = sec * 4 + (tid >> 6);
        for (int k = gwave; k < nz; k += LSNF_UNFOLD_SECTIONS * 4) {
            double t1 = 0.0, t2 = 0.0;
            for (int n = lane; n < nz; n += 64) {
                const double wkn = (double)P[P_W][k * nz + n];
                t1 += wkn * (double)F[fl.dWa + k * nz + n];
                t2 += wkn * (double)F[fl.dca + n];
            }
            t1 = wave_sum(t1); t2 = wave_sum(t2);
            if (lane == 0) {
                const double e = ea[k], b = (double)P[P_AB][k];
                if (Gp[P_AB]) Gp[P_AB][k] = (float)(e * t2);
                if (Gp[P_ALOGS]) Gp[P_ALOGS][k] = (float)(3.0 * e * (t1 + b * t2) + 3.0 * Gtot);
            }
        }
    }
    if (Gp[P_W])
        for (int idx = gtid; idx < nz * nz; idx += gstride) {
            const int k = idx / nz, n = idx % nz;
            const double e = ea[k], b = (double)P[P_AB][k];
            Gp[P_W][idx] = (float)(e * (double)F[fl.dWa + idx] + b * e * (double)F[fl.dca + n] + Gtot * (double)winv[n * nz + k]);
        }
    // ---- fc_1 / fc_2:  W' = W diag(e), c = b*e
    for (int layer = 0; layer < 2; ++layer) {
        const int iw = layer ? P_W2 : P_W1, ib = layer ? P_B2 : P_B1, il = layer ? P_LOGS2 : P_LOGS1;
        const int rows = layer ? w : half, oW = layer ? fl.dW2 : fl.dW1, oc = layer ? fl.dc2 : fl.dc1;
        const double* ev = layer ? e2v : e1v;
        if (sec == layer)
            for (int n = tid; n < w; n += 256) {
                const double e = ev[n], b = (double)P[ib][n], dc = (double)F[oc + n];
                double u = 0.0;
                for (int k = 0; k < rows; ++k) u += (double)P[iw][k * w + n] * (double)F[oW + k * w + n];
                if (Gp[ib]) Gp[ib][n] = (float)(dc * e);
                if (Gp[il]) Gp[il][n] = (float)(3.0 * e * (u + b * dc));
            }
        if (Gp[iw])
            for (int idx = gtid; idx < rows * w; idx += gstride) {
                const int n = idx % w;
                Gp[iw][idx] = (float)((double)F[oW + idx] * ev[n]);
            }
    }
    // ---- fc_zeros: affine: column c = 2f + which (shift / pre-sigmoid interleaved, model.py:411-413);
    //      additive (model.py:407-408): column c = f, shift only
    if (sec == 2)
        for (int c = tid; c < n3; c += 256) {
            const int f = coupling ? (c >> 1) : c, which = coupling ? (c & 1) : 0;
            const int oW = which ? fl.dW3p : fl.dW3s, oc = which ? fl.dc3p : fl.dc3s;
            const double e = e3v[c], b = (double)P[P_B3][c], dc = (double)F[oc + f];
            double u = 0.0;
            for (int k = 0; k < w; ++k) u += (double)P[P_W3][k * n3 + c] * (double)F[oW + k * half + f];
            if (Gp[P_B3]) Gp[P_B3][c] = (float)(dc * e);
            if (Gp[P_LOGS3]) Gp[P_LOGS3][c] = (float)(3.0 * e * (u + b * dc));
        }
    if (Gp[P_W3])
        for (int idx = gtid; idx < w * n3; idx += gstride) {
            const int k = idx / n3, c = idx % n3, f = coupling ? (c >> 1) : c, which = coupling ? (c & 1) : 0;
            const int oW = which ? fl.dW3p : fl.dW3s;
            Gp[P_W3][idx] = (float)((double)F[oW + k * half + f] * e3v[c]);
        }
}
}  // namespace

hipError_t lsnf_launch_backward_params(const LsnfGeo& g, const float* plan, const float* const* params_host,
                                       float* const* grads_host, int B, const float* z_in, const float* z_out,
                                       const float* z_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                                       float ll_scale, float* g_z_in, float* workspace, int vec4, int small_batch,
                                       hipStream_t stream) {
    const LsnfFoldLayout fl = lsnf_fold_layout(g.nz, g.width);
    float* gl_total = workspace;
    float* fold = workspace + 4;
    float* dump = fold + (size_t)g.depth * fl.per_block;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(float) * (4 + (size_t)g.depth * fl.per_block), stream);
    if (e != hipSuccess) return e;
    e = small_batch
        ? lsnf_launch_small_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4, stream,
                                       nullptr, nullptr, dump, gl_total)
        : lsnf_launch_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, dump, gl_total, vec4, stream);
    if (e != hipSuccess) return e;
    TnArgs t;
    t.z_in = z_in; t.z_out = z_out; t.z_saved = z_saved; t.dump = dump; t.fold = fold;
    t.B = B; t.nz = g.nz; t.half = g.half; t.width = g.width; t.depth = g.depth;
    t.chunk = B >= 16384 ? 1024 : 128;               // samples per workgroup (multiple of 16)
    const unsigned chunks = (unsigned)((B + t.chunk - 1) / t.chunk);
    hipLaunchKernelGGL(lsnf_tn_gemm_kernel, dim3(g.depth * 5, chunks), dim3(256), 0, stream, t);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    LsnfParamPtrs pp; LsnfGradPtrs gp;
    for (int i = 0; i < LSNF_MAX_DEPTH * 12; ++i) {
        pp.p[i] = i < g.depth * 12 ? params_host[i] : nullptr;
        gp.p[i] = i < g.depth * 12 ? grads_host[i] : nullptr;
    }
    hipLaunchKernelGGL(lsnf_unfold_kernel, dim3(g.depth, LSNF_UNFOLD_SECTIONS), dim3(256), 0, stream, pp, gp, (const float*)fold,
                       (const float*)gl_total, plan + g.off_winv, g.nz, g.width, g.coupling);
    return hipGetLastError();
}
