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
#include <stdlib.h>
#include "lsnf_layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

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

hipError_t lsnf_launch_small3_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                         const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                                         float ll_scale, float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                         float* dump, float* gl_total);
hipError_t lsnf_launch_backward3_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                   const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                   float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                   float* dump, float* gl_total, int dump_tiled);

// lsnf_params3.hip: the same contraction on the bf16 matrix pipe (large batches); hipErrorInvalidValue = not covered
hipError_t lsnf_launch_contract_x3(const float* z_in, const float* z_out, const float* z_saved, const float* dump, float* fold,
                                   int B, int nz, int half, int width, int depth, int chunk_override, int g_tiled, const int* h_tag,
                                   hipStream_t stream);
bool lsnf_contract_x3_covers(int B, int nz, int half, int width, const float* z_in, const float* z_out, const float* z_saved);

namespace {

struct TnArgs {
    const float* z_in; const float* z_out; const float* z_saved;
    const float* dump; float* fold;
    int B, nz, half, width, depth, chunk;
    int abl;                           // timing experiments only (tools/): 1 no atomics, 2 no MFMA, 4 no global loads after the first stage
};

// Output tiles (kt, nt) of a task (KT x NT tiles of 32 x 32) over the 4 waves, so that every wave (SIMD) has matrix work for every
// task shape: KT >= 3: wave = k-tile, all n-tiles (dWa 4x4); KT = 2: wave = (k-tile, n-tile parity) (the four 2x2 tasks of
// f_width 64 were running on two waves); KT = 1: wave = n-tile.  The wave with kt == 0 also sums its G columns (bias gradients).
__device__ __forceinline__ void tn_wave_tiles(int wave, int KT, int NT, int& kt, unsigned& ntmask) {
    if (KT >= 3) { kt = wave; ntmask = wave < KT ? ((1u << NT) - 1u) : 0u; }
    else if (KT == 2) {
        kt = wave & 1;
        const int q = wave >> 1;
        ntmask = 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < NT && (t & 1) == q) ntmask |= 1u << t;
    } else { kt = 0; ntmask = wave < NT ? (1u << wave) : 0u; }
}

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
    const int NT = (N + 31) / 32;
    int kt; unsigned own;
    tn_wave_tiles(__builtin_amdgcn_readfirstlane(wave), (K + 31) / 32, NT, kt, own);
    if (own == 0u) return;                       // wave-uniform
    const int kcol = 32 * kt + i;
    const bool kok = kcol < K;
    f32x16 acc[4];
    float csum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { csum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; }
    const int m_begin = blockIdx.y * a.chunk;
    const int m_end = min(a.B, m_begin + a.chunk);
    // U sample pairs per trip, all loads of a trip issued before its MFMAs -- and the NEXT trip's loads before them too (two
    // register sets): a workgroup's trips are a dependent chain, so without that every trip pays one memory latency
    // (B = 100: seven trips, 23 us; with the loads one trip ahead ~8 us).  One workgroup per task up to 128 rows: no atomics
    // race, the sums are bit-reproducible at the reference's batch size.
    constexpr int U = 8;
    int ncol[4]; bool nok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ncol[t] = 32 * t + i; nok[t] = ((own >> t) & 1u) && ncol[t] < N; }
    auto load_trip = [&](int m0, float (&av)[U], float (&gv)[U][4]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 2 * u + kk;
            const bool mok = m < m_end;
            av[u] = (mok && kok) ? A[(size_t)m * lda + kcol] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) gv[u][t] = (mok && nok[t]) ? G[(size_t)m * ldg + ncol[t]] : 0.f;
        }
    };
    auto mma_trip = [&](const float (&av)[U], const float (&gv)[U][4]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if ((own >> t) & 1u) {
                    csum[t] += gv[u][t];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], gv[u][t], acc[t], 0, 0, 0);
                }
            }
        }
    };
    float av0[U], gv0[U][4], av1[U], gv1[U][4];
    load_trip(m_begin, av0, gv0);
    for (int m0 = m_begin; m0 < m_end; m0 += 4 * U) {      // two trips per iteration: the register sets alternate statically
        load_trip(m0 + 2 * U, av1, gv1);                       // (rows past m_end load zeros)
        mma_trip(av0, gv0);
        if (m0 + 2 * U < m_end) {
            load_trip(m0 + 4 * U, av0, gv0);
            mma_trip(av1, gv1);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if ((own >> t) & 1u) {
            const int ncol = 32 * t + i;
            if (ncol < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (krow < K) atomicAdd(&C[(size_t)krow * N + ncol], acc[t][r]);
                }
            }
            if (kt == 0) {     // column sums of G (bias gradients), once per task and n-tile
                const float s = csum[t] + __shfl_xor(csum[t], 32, 64);
                if (kk == 0 && ncol < N) atomicAdd(&cs[ncol], s);
            }
        }
    }
}

// The same contraction for LARGE batches, staged through LDS.  lsnf_tn_gemm_kernel above loads one float per lane and sample
// (40 dword loads per 16 samples and wave, the G tiles four times per workgroup -- once by every wave): 489 us at B = 65 536,
// where the contraction's HBM traffic (1.0 GB) allows ~0.2 ms and its fp32-MFMA work ~0.15 ms.  Here a workgroup stages
// TN_S samples of A (S x K) and G (S x N) with 16-byte row loads (each element read ONCE per workgroup), double-buffered,
// and every wave reads its MFMA operands from LDS (one conflict-free ds_read_b32 per operand: lanes run along the
// feature index).  Same tasks, same atomics into the folded buffer.
#ifndef TN_S
#define TN_S 16                         // samples per stage (32 KiB of LDS per workgroup: five workgroups per CU overlap each other's stage turnover)
#endif
template <int VW>                       // vector width of the row loads: 4 / 2 / 1 floats (alignment of the task's A and G rows)
__device__ __forceinline__ void tn_stage_load(float (&regs)[TN_S * 128 / 256], const float* __restrict__ src, int ld, int ncols, int m0, int m_end,
                                              int tid) {
    // tile = TN_S rows x 128 columns (columns >= ncols are zero); thread `tid` owns elements e = (tid + 256*j) * VW .. +VW-1
    constexpr int PER = TN_S * 128 / 256;          // floats per thread
#pragma unroll
    for (int j = 0; j < PER / VW; ++j) {
        const int e = (tid + 256 * j) * VW, r = e >> 7, c = e & 127, m = m0 + r;
        const bool ok = m < m_end && c < ncols;
        const float* q = src + (size_t)m * ld + c;
        if constexpr (VW == 4) {
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4v*>(q);      // (ncols % 4 == 0 on this path: a 4-group is all in or all out)
            regs[4 * j] = v[0]; regs[4 * j + 1] = v[1]; regs[4 * j + 2] = v[2]; regs[4 * j + 3] = v[3];
        } else if constexpr (VW == 2) {
            f32x2v v = {0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x2v*>(q);
            regs[2 * j] = v[0]; regs[2 * j + 1] = v[1];
        } else {
            regs[j] = ok ? *q : 0.0f;
        }
    }
}
template <int VW>
__device__ __forceinline__ void tn_stage_store(const float (&regs)[TN_S * 128 / 256], float* lds, int tid) {
    constexpr int PER = TN_S * 128 / 256;
#pragma unroll
    for (int j = 0; j < PER / VW; ++j) {
        const int e = (tid + 256 * j) * VW;
        if constexpr (VW == 4) { f32x4v v = {regs[4 * j], regs[4 * j + 1], regs[4 * j + 2], regs[4 * j + 3]}; *reinterpret_cast<f32x4v*>(lds + e) = v; }
        else if constexpr (VW == 2) { f32x2v v = {regs[2 * j], regs[2 * j + 1]}; *reinterpret_cast<f32x2v*>(lds + e) = v; }
        else lds[e] = regs[j];
    }
}
template <int VW>
__global__ __launch_bounds__(256) void lsnf_tn_gemm_lds_kernel(const TnArgs a) {
    const int task = blockIdx.x, blk = task / 5, which = task % 5;
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const LsnfFoldLayout fl = lsnf_fold_layout(a.nz, a.width);
    const float* dmp = a.dump + (size_t)blk * dl.per_block;
    float* fold = a.fold + (size_t)blk * fl.per_block;
    const float* yblk = (blk == a.depth - 1) ? a.z_out : a.z_saved + (size_t)blk * a.B * a.nz;
    const float* xblk = (blk == 0) ? a.z_in : a.z_saved + (size_t)(blk - 1) * a.B * a.nz;
    const float *A, *G; int lda, ldg, K, N; float *C, *cs;
    switch (which) {
    case 0: A = xblk; lda = a.nz; K = a.nz; G = dmp + dl.off_gv; ldg = a.nz; N = a.nz; C = fold + fl.dWa; cs = fold + fl.dca; break;
    case 1: A = yblk; lda = a.nz; K = a.half; G = dmp + dl.off_ga1; ldg = a.width; N = a.width; C = fold + fl.dW1; cs = fold + fl.dc1; break;
    case 2: A = dmp + dl.off_h1; lda = a.width; K = a.width; G = dmp + dl.off_ga2; ldg = a.width; N = a.width; C = fold + fl.dW2; cs = fold + fl.dc2; break;
    case 3: A = dmp + dl.off_h2; lda = a.width; K = a.width; G = dmp + dl.off_gt; ldg = a.half; N = a.half; C = fold + fl.dW3s; cs = fold + fl.dc3s; break;
    default: A = dmp + dl.off_h2; lda = a.width; K = a.width; G = dmp + dl.off_gp; ldg = a.half; N = a.half; C = fold + fl.dW3p; cs = fold + fl.dc3p; break;
    }
    __shared__ __attribute__((aligned(16))) float sA[2][TN_S * 128], sG[2][TN_S * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kk = lane >> 5;
    const int NT = (N + 31) / 32;
    int kt; unsigned own;
    tn_wave_tiles(__builtin_amdgcn_readfirstlane(wave), (K + 31) / 32, NT, kt, own);
    const bool kwave = own != 0u;                // this wave has output tiles
    f32x16 acc[4];
    float csum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { csum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; }
    const int m_begin = blockIdx.y * a.chunk, m_end = min(a.B, m_begin + a.chunk);
    float ra[TN_S * 128 / 256], rg[TN_S * 128 / 256];
    tn_stage_load<VW>(ra, A, lda, K, m_begin, m_end, tid);
    tn_stage_load<VW>(rg, G, ldg, N, m_begin, m_end, tid);
    tn_stage_store<VW>(ra, sA[0], tid);
    tn_stage_store<VW>(rg, sG[0], tid);
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += TN_S) {
        const bool more = m0 + TN_S < m_end;      // workgroup-uniform
        if (more && !(a.abl & 4)) {                // next stage's rows in flight under this stage's MFMAs
            tn_stage_load<VW>(ra, A, lda, K, m0 + TN_S, m_end, tid);
            tn_stage_load<VW>(rg, G, ldg, N, m0 + TN_S, m_end, tid);
        }
        if (kwave) {
            const float* pa = sA[cur] + kk * 128 + 32 * kt + i;
            const float* pg = sG[cur] + kk * 128 + i;
#pragma unroll
            for (int u = 0; u < TN_S / 2; ++u) {
                const float av = pa[u * 256];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((own >> t) & 1u) {
                        const float gv = pg[u * 256 + 32 * t];
                        csum[t] += gv;
                        if (!(a.abl & 2)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            tn_stage_store<VW>(ra, sA[cur ^ 1], tid);
            tn_stage_store<VW>(rg, sG[cur ^ 1], tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (!kwave) return;
    if ((a.abl & 1) && blockIdx.y != 0) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if ((own >> t) & 1u) {
            const int ncol = 32 * t + i;
            if (ncol < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (krow < K) atomicAdd(&C[(size_t)krow * N + ncol], acc[t][r]);
                }
            }
            if (kt == 0) {     // column sums of G (bias gradients), once per task and n-tile
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
    // column reductions u_n = sum_k W_kn dW'_kn: all 256 threads of the section that owns the layer, lanes along n (coalesced),
    // the rows split over 256 / 128 = 2 .. 256 / 32 = 8 thread groups, partial sums combined through LDS
    // (one thread per column walking all rows was the critical path of the whole kernel: 41 us, now ~10)
    __shared__ double part[256];
    for (int layer = 0; layer < 3; ++layer) {        // 0: fc_1, 1: fc_2, 2: fc_zeros
        const int iw = layer == 0 ? P_W1 : (layer == 1 ? P_W2 : P_W3), ib = layer == 0 ? P_B1 : (layer == 1 ? P_B2 : P_B3);
        const int il = layer == 0 ? P_LOGS1 : (layer == 1 ? P_LOGS2 : P_LOGS3);
        const int rows = layer == 0 ? half : w, cols = layer == 2 ? n3 : w;
        const double* ev = layer == 0 ? e1v : (layer == 1 ? e2v : e3v);
        if (sec == layer) {                           // section-uniform
            const int cpad = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128), groups = 256 / cpad;
            const int n = tid % cpad, q = tid / cpad;
            double u = 0.0;
            if (n < cols) {
                // (loads of four rows in flight per trip: a dependent load -> fma chain per row costs one memory latency each)
                const float* pw; const float* pf; int sw, sf;
                if (layer < 2) { pw = P[iw] + n; sw = w; pf = F + (layer ? fl.dW2 : fl.dW1) + n; sf = w; }
                else {                                // fc_zeros: column c = 2f + which (affine, model.py:411-413) or c = f (additive)
                    const int f = coupling ? (n >> 1) : n, which = coupling ? (n & 1) : 0;
                    pw = P[iw] + n; sw = n3; pf = F + (which ? fl.dW3p : fl.dW3s) + f; sf = half;
                }
                int k = q;
                for (; k + 3 * groups < rows; k += 4 * groups) {
                    float a[4], b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = pw[(size_t)(k + j * groups) * sw]; b[j] = pf[(size_t)(k + j * groups) * sf]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) u += (double)a[j] * (double)b[j];
                }
                for (; k < rows; k += groups) u += (double)pw[(size_t)k * sw] * (double)pf[(size_t)k * sf];
            }
            part[tid] = u;
            __syncthreads();
            if (q == 0 && n < cols) {
                for (int j = 1; j < groups; ++j) u += part[j * cpad + n];
                double dc;
                if (layer < 2) dc = (double)F[(layer ? fl.dc2 : fl.dc1) + n];
                else { const int f = coupling ? (n >> 1) : n; dc = (double)F[((coupling && (n & 1)) ? fl.dc3p : fl.dc3s) + f]; }
                const double e = ev[n], b = (double)P[ib][n];
                if (Gp[ib]) Gp[ib][n] = (float)(dc * e);
                if (Gp[il]) Gp[il][n] = (float)(3.0 * e * (u + b * dc));
            }
        }
        if (layer < 2 && Gp[iw]) {
            const int oW = layer ? fl.dW2 : fl.dW1;
            for (int idx = gtid; idx < rows * w; idx += gstride) {
                const int n = idx % w;
                Gp[iw][idx] = (float)((double)F[oW + idx] * ev[n]);
            }
        }
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
                                       hipStream_t stream, const float* act_saved) {
    const LsnfFoldLayout fl = lsnf_fold_layout(g.nz, g.width);
    float* gl_total = workspace;
    float* fold = workspace + 4;
    float* dump = fold + (size_t)g.depth * fl.per_block;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(float) * (4 + (size_t)g.depth * fl.per_block), stream);
    if (e != hipSuccess) return e;
    // Fast path (act_saved given, bf16x3-family math): the forward of this evaluation kept the activation stash and wrote h1 / h2
    // into this workspace's dump (lsnf_forward(params_workspace)); the backward FROM THE STASH, on the bf16 matrix pipe, adds
    // g_v, g_a1, g_a2, g_t, g_p -- no recomputation of the coupling MLP (1.0x instead of 1.5x the forward's matrix work, at
    // 2.6x the matrix rate).  Otherwise: the recomputing fp32-MFMA backward writes all seven tensors itself.
    // Large batches on the fast path (bf16x3-family math: act_saved given): the contraction runs on the bf16 matrix pipe, operands read
    // once (lsnf_params3.hip; LSNF_TN_X3=0 keeps the fp32-MFMA kernels), and the throughput backward then writes its g arrays in the
    // tiled form (whole 1 KiB stores instead of 16 rows x 64 bytes; g_v as its first half only: the second half is g_t)
    static const bool knob_plain = getenv("LSNF_TN_PLAIN") != nullptr;
    const bool use_x3 = act_saved && lsnf_contract_x3_covers(B, g.nz, g.half, g.width, z_in, z_out, z_saved);
    const int g_tiled = (use_x3 && !small_batch && lsnf_dump_can_tile(g.nz, g.width)) ? 1 : 0;
    if (act_saved) {
        e = small_batch
            ? lsnf_launch_small3_backward_z(g, plan, B, z_out, z_saved, act_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4, stream,
                                            nullptr, dump, gl_total)
            : lsnf_launch_backward3_z(g, plan, B, z_out, z_saved, act_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4, stream,
                                      nullptr, dump, gl_total, g_tiled);
    } else
    e = small_batch
        ? lsnf_launch_small_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4, stream,
                                       nullptr, nullptr, dump, gl_total)
        : lsnf_launch_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, dump, gl_total, vec4, stream);
    if (e != hipSuccess) return e;
    TnArgs t;
    t.z_in = z_in; t.z_out = z_out; t.z_saved = z_saved; t.dump = dump; t.fold = fold;
    t.B = B; t.nz = g.nz; t.half = g.half; t.width = g.width; t.depth = g.depth;
    t.chunk = B >= 16384 ? 512 : 128;                // samples per workgroup (multiple of 16; 512: 3 200 workgroups at B = 65 536 -- measured 620 us for the whole call against 672 at 1 024 and 760 at 2 048)
    // experiment knobs of tools/tn_probe.py (read once per process): ablation switches, samples per workgroup, plain kernel
    static const int knob_abl = [] { const char* e = getenv("LSNF_TN_ABL"); return e ? atoi(e) : 0; }();
    static const int knob_chunk = [] { const char* e = getenv("LSNF_TN_CHUNK"); return e ? atoi(e) : 0; }();
    t.abl = knob_abl;
    if (knob_chunk > 0) t.chunk = knob_chunk;
    const unsigned chunks = (unsigned)((B + t.chunk - 1) / t.chunk);
    e = hipErrorInvalidValue;
    if (use_x3)
        e = lsnf_launch_contract_x3(z_in, z_out, z_saved, dump, fold, B, g.nz, g.half, g.width, g.depth, knob_chunk, g_tiled,
                                    reinterpret_cast<const int*>(workspace + lsnf_params_workspace_tag(g.nz, g.width, g.depth, B)), stream);
    if (e == hipErrorInvalidValue && g_tiled) return hipErrorUnknown;        // (cannot happen: lsnf_contract_x3_covers said yes)
    if (e == hipSuccess) {
    } else if (e != hipErrorInvalidValue) {
        return e;
    } else if (B >= 4096 && !knob_plain) {
        // every row the tasks read starts 16-byte aligned iff nz, width and half are multiples of 4 (z tensors: the caller's
        // alignment is folded into vec4; the dump rows start at 16-byte aligned offsets of the 16-byte aligned workspace)
        const bool a4 = vec4 == 4 && g.nz % 4 == 0 && g.width % 4 == 0 && g.half % 4 == 0;
        const bool a2 = vec4 >= 2 && g.nz % 2 == 0 && g.width % 2 == 0 && g.half % 2 == 0;
        if (a4) hipLaunchKernelGGL(lsnf_tn_gemm_lds_kernel<4>, dim3(g.depth * 5, chunks), dim3(256), 0, stream, t);
        else if (a2) hipLaunchKernelGGL(lsnf_tn_gemm_lds_kernel<2>, dim3(g.depth * 5, chunks), dim3(256), 0, stream, t);
        else hipLaunchKernelGGL(lsnf_tn_gemm_lds_kernel<1>, dim3(g.depth * 5, chunks), dim3(256), 0, stream, t);
    } else
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
