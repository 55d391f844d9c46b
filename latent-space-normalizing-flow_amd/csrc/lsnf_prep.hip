// lsnf_prep.hip -- batch-independent weight preparation (once per parameter update):
//   (1) lsnf_gj_kernel   : float64 Gauss-Jordan with partial pivoting of every block's 1x1-conv
//                          matrix W in LDS -> log|det W| (reference model.py:182 computes
//                          log|det(w.double())| on every call) and W^-1 (model.py:193).
//   (2) lsnf_pack_kernel : folds exp(3*logs) of the three actnorms into the adjacent matrices
//                          (model.py:264-268), de-interleaves fc_zeros' shift/scale columns
//                          (model.py:411-413, +2 folded into the bias), zero-pads to 32-tiles and
//                          writes everything in MFMA-fragment order (lsnf_layout.h), for the
//                          forward, reverse and backward streams.
// All folding arithmetic is float64, rounded to float32 once.
#include <hip/hip_runtime.h>
#include "lsnf_layout.h"

struct LsnfParamPtrs {
    const float* p[LSNF_MAX_DEPTH * 12];
};

enum { P_AB = 0, P_ALOGS, P_W, P_W1, P_B1, P_LOGS1, P_W2, P_B2, P_LOGS2, P_W3, P_B3, P_LOGS3 };

// scratch layout (doubles): per block [nz*nz inverse][8: logabsdet, sum3logs, ...]
__host__ __device__ static inline size_t scratch_block_doubles(int nz) { return (size_t)nz * nz + 8; }

// ---------------------------------------------------------------------------------------------
// (1) Gauss-Jordan: one workgroup per block, matrix resident in LDS as double [n][n+1].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsnf_gj_kernel(LsnfParamPtrs pp, int nz, double* scratch) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = nz, ld = n + 1;         // odd leading dimension: column walks are bank-conflict free
    double* A = sm;                       // n * ld   in-place inverse
    double* prow = A + (size_t)n * ld;    // n   scaled pivot row of the current step
    double* colv = prow + n;              // n   eliminated column of the current step
    int* perm = (int*)(colv + n);         // n   pivot row chosen at step c
    int* cmap = perm + n;                 // n   final column un-permutation
    int* pivot = cmap + n;                // 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ry = tid >> 4, cx = tid & 15;            // 16 x 16 thread grid over the matrix
    const int blk = blockIdx.x;
    const float* W = pp.p[blk * 12 + P_W];
    for (int r = ry; r < n; r += 16)
        for (int j = cx; j < n; j += 16) A[r * ld + j] = (double)W[r * n + j];
    __syncthreads();
    double logabs = 0.0;                  // meaningful in thread 0
    for (int c = 0; c < n; ++c) {
        // (1) partial pivoting: wave 0 scans rows c..n-1 of column c, shuffle arg-max (ties -> lowest row)
        if (wave == 0) {
            double best = -1.0; int bi = c;
            for (int r = c + lane; r < n; r += 64) {
                const double v = fabs(A[r * ld + c]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { *pivot = bi; perm[c] = bi; }
        }
        __syncthreads();
        const int p = *pivot;
        // (2) read phase: rows c and p (to be exchanged) and column c
        const double pv = A[p * ld + c];
        double a_c = 0.0, a_p = 0.0, cv = 0.0;
        if (tid < n) { a_c = A[c * ld + tid]; a_p = A[p * ld + tid]; }
        else if (tid < 2 * n) { const int r = tid - n; cv = A[r * ld + c]; }
        __syncthreads();
        // (3) write phase: exchange, scale the pivot row (pivot slot becomes 1/pv), publish row / column
        const double ipv = 1.0 / pv;
        if (tid == 0) logabs += log(fabs(pv));
        if (tid < n) {
            const int j = tid;
            const double pr = (j == c ? 1.0 : a_p) * ipv;
            if (p != c) A[p * ld + j] = a_c;
            A[c * ld + j] = pr;
            prow[j] = pr;
        } else if (tid < 2 * n) {
            // multipliers of the elimination = column c AFTER the exchange: rows other than c, p keep theirs;
            // row p now holds the old row c, whose entry old A[c][c] was read by the thread of r == c.
            const int r = tid - n;
            if (r == c) { if (p != c) colv[p] = cv; }
            else if (r != p) colv[r] = cv;
        }
        __syncthreads();
        // (4) eliminate column c from every row but c: 8 x 8 elements per thread, fully unrolled so that the
        //     LDS reads of a thread are all in flight together (one workgroup per CU: latency, not bandwidth)
        {
            double pr[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { const int j = cx + 16 * jj; pr[jj] = (j < n) ? prow[j] : 0.0; }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = ry + 16 * rr;
                if (r < n && r != c) {
                    const double f = colv[r];
                    double av[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) { const int j = cx + 16 * jj; av[jj] = (j < n && j != c) ? A[r * ld + j] : 0.0; }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) { const int j = cx + 16 * jj; if (j < n) A[r * ld + j] = av[jj] - f * pr[jj]; }
                }
            }
        }
        __syncthreads();
    }
    // undo the row interchanges as column interchanges (reverse order), composed into one map by thread 0
    if (tid == 0) {
        for (int j = 0; j < n; ++j) cmap[j] = j;
        for (int c = n - 1; c >= 0; --c) {
            const int p = perm[c];
            if (p != c) { const int t = cmap[c]; cmap[c] = cmap[p]; cmap[p] = t; }
        }
    }
    __syncthreads();
    double* out = scratch + (size_t)blk * scratch_block_doubles(nz);
    // after the swaps, position j holds what the un-swapped matrix has in column cmap[j]
    for (int r = ry; r < n; r += 16)
        for (int j = cx; j < n; j += 16) out[r * n + j] = A[r * ld + cmap[j]];
    if (tid == 0) {
        const float* logs = pp.p[blk * 12 + P_ALOGS];
        // reference: torch.sum(logs * 3) in fp32 (model.py:264,273); each term logs*3 is rounded
        // to fp32 first, then summed -- we sum those fp32 terms in double and round once.
        double s3 = 0.0;
        for (int k = 0; k < n; ++k) s3 += (double)(logs[k] * 3.0f);
        out[(size_t)n * n + 0] = logabs;
        out[(size_t)n * n + 1] = s3;
    }
}

// ---------------------------------------------------------------------------------------------
// (2) packing
// ---------------------------------------------------------------------------------------------
struct SplitIdx { int nat; bool ok; };
// split-pad coordinate c in [0, 64*HT) -> natural latent column
__device__ static inline SplitIdx split_nat(int c, int HT, int half) {
    const int t = c / 32, o = c % 32;
    const int hh = t / HT, f = 32 * (t % HT) + o;
    SplitIdx s; s.ok = f < half; s.nat = hh * half + f; return s;
}

__device__ static inline double e3(const float* logs, int i) { return exp((double)(logs[i] * 3.0f)); }

// Forward-stage matrices in padded coordinates.  stage 1..4 ; returns M[k][n].
__device__ static double fwd_mat(const LsnfGeo& g, const float* const* P, int stage, int k, int n) {
    const int half = g.half, w = g.width, nz = g.nz, HT = g.HT;
    switch (stage) {
    case 1: {   // Wa = diag(exp(3 logs_a)) W
        const SplitIdx sk = split_nat(k, HT, half), sn = split_nat(n, HT, half);
        if (!sk.ok || !sn.ok) return 0.0;
        return e3(P[P_ALOGS], sk.nat) * (double)P[P_W][sk.nat * nz + sn.nat];
    }
    case 2:     // W1' = W1 diag(exp(3 logs1)) ; k = first-half feature, n = hidden unit
        if (k >= half || n >= w) return 0.0;
        return (double)P[P_W1][k * w + n] * e3(P[P_LOGS1], n);
    case 3:
        if (k >= w || n >= w) return 0.0;
        return (double)P[P_W2][k * w + n] * e3(P[P_LOGS2], n);
    default: {  // 4: n in [0,32HT) -> shift column 2f ; [32HT,64HT) -> pre-sigmoid column 2f+1
        const int which = n / (32 * HT), f = n % (32 * HT);
        if (k >= w || f >= half) return 0.0;
        const int col = 2 * f + which;
        return (double)P[P_W3][k * nz + col] * e3(P[P_LOGS3], col);
    }
    }
}

__device__ static double fwd_bias(const LsnfGeo& g, const float* const* P, int stage, int n) {
    const int half = g.half, w = g.width, nz = g.nz, HT = g.HT;
    switch (stage) {
    case 1: {   // ca = (b_a * exp(3 logs_a)) @ W
        const SplitIdx sn = split_nat(n, HT, half);
        if (!sn.ok) return 0.0;
        double s = 0.0;
        for (int k = 0; k < nz; ++k) s += (double)P[P_AB][k] * e3(P[P_ALOGS], k) * (double)P[P_W][k * nz + sn.nat];
        return s;
    }
    case 2: return n < w ? (double)P[P_B1][n] * e3(P[P_LOGS1], n) : 0.0;
    case 3: return n < w ? (double)P[P_B2][n] * e3(P[P_LOGS2], n) : 0.0;
    default: {
        const int which = n / (32 * HT), f = n % (32 * HT);
        if (f >= half) return which ? 40.0 : 0.0;   // padded scale lane: sigmoid(40) == 1, log == 0
        const int col = 2 * f + which;
        return (double)P[P_B3][col] * e3(P[P_LOGS3], col) + (which ? 2.0 : 0.0);   // "+ 2." model.py:413
    }
    }
}

// decode a float index inside a stage's panel stream: stage has KT k-tiles per panel
__device__ static inline void frag_decode(int q, int KT, int* k, int* n) {
    const int per_panel = KT * LSNF_FRAG_FLOATS;
    const int nt = q / per_panel; int r = q % per_panel;
    const int kt = r / LSNF_FRAG_FLOATS; r %= LSNF_FRAG_FLOATS;
    const int gq = r / 256; r %= 256;
    const int lane = r / 4, j = r % 4;
    *k = 32 * kt + 8 * gq + 4 * (lane >> 5) + j;
    *n = 32 * nt + (lane & 31);
}

// bias block order inside a stage: index = nt*32 + h*16 + r  <->  feature 32*nt + o(r,h)
__device__ static inline int bias_feature(int i) {
    const int nt = i / 32, h = (i % 32) / 16, r = i % 16;
    return 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h;
}

__global__ __launch_bounds__(256) void lsnf_pack_kernel(LsnfParamPtrs pp, LsnfGeo g, const double* scratch, float* plan) {
    const int blk = blockIdx.y;
    const float* const* P = &pp.p[blk * 12];
    const int HT = g.HT, WT = g.WT, NZT = g.NZT;
    const double* sb = scratch + (size_t)blk * scratch_block_doubles(g.nz);
    const int nz = g.nz;
    // region sizes for this block
    const int n_fc = g.fwd_const_floats, n_fp = g.fwd_block_floats;
    const int n_ic = g.inv_const_floats, n_ip = g.inv_block_floats;
    const int n_bp = g.bwd_block_floats, n_wi = nz * nz;
    const int total = n_fc + n_fp + n_ic + n_ip + n_bp + n_wi;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int q = idx;
        if (q < n_fc) {                     // ---- forward constants
            float* dst = plan + g.off_fwd_const + (size_t)blk * n_fc;
            const int nb = 32 * g.fwd_panels;
            double val = 0.0;
            if (q < nb) {
                int i = q, stage;
                if (i < 32 * NZT) stage = 1;
                else if ((i -= 32 * NZT) < 32 * WT) stage = 2;
                else if ((i -= 32 * WT) < 32 * WT) stage = 3;
                else { i -= 32 * WT; stage = 4; }
                val = fwd_bias(g, P, stage, bias_feature(i));
            } else if (q == nb + 0) val = sb[(size_t)nz * nz + 1];      // sum(3*logs_a)
            else if (q == nb + 1) val = sb[(size_t)nz * nz + 0];        // log|det W|
            dst[q] = (float)val;
            continue;
        }
        q -= n_fc;
        if (q < n_fp) {                     // ---- forward panels
            float* dst = plan + g.off_fwd_panels + (size_t)blk * n_fp;
            int r = q, k, n, stage, KT;
            const int s1 = LSNF_FRAG_FLOATS * NZT * NZT, s2 = LSNF_FRAG_FLOATS * WT * HT, s3 = LSNF_FRAG_FLOATS * WT * WT;
            if (r < s1) { stage = 1; KT = NZT; }
            else if ((r -= s1) < s2) { stage = 2; KT = HT; }
            else if ((r -= s2) < s3) { stage = 3; KT = WT; }
            else { r -= s3; stage = 4; KT = WT; }
            frag_decode(r, KT, &k, &n);
            dst[q] = (float)fwd_mat(g, P, stage, k, n);
            continue;
        }
        q -= n_fp;
        if (q < n_ic) {                     // ---- inverse constants: bias = -b_a  (model.py:246)
            float* dst = plan + g.off_inv_const + (size_t)blk * n_ic;
            const SplitIdx sn = split_nat(bias_feature(q), HT, g.half);
            dst[q] = sn.ok ? -P[P_AB][sn.nat] : 0.0f;
            continue;
        }
        q -= n_ic;
        if (q < n_ip) {                     // ---- inverse panel I1: Winv' = Winv diag(exp(-3 logs_a))
            float* dst = plan + g.off_inv_panels + (size_t)blk * n_ip;
            int k, n;
            frag_decode(q, NZT, &k, &n);
            const SplitIdx sk = split_nat(k, HT, g.half), sn = split_nat(n, HT, g.half);
            double val = 0.0;
            if (sk.ok && sn.ok) val = sb[(size_t)sk.nat * nz + sn.nat] * exp(-(double)(P[P_ALOGS][sn.nat] * 3.0f));
            dst[q] = (float)val;
            continue;
        }
        q -= n_ip;
        if (q < n_bp) {                     // ---- backward panels: transposes of the forward matrices
            float* dst = plan + g.off_bwd_panels + (size_t)blk * n_bp;
            int r = q, k, n;
            const int b4 = LSNF_FRAG_FLOATS * WT * 2 * HT, b3 = LSNF_FRAG_FLOATS * WT * WT, b2 = LSNF_FRAG_FLOATS * HT * WT;
            double val;
            if (r < b4) { frag_decode(r, 2 * HT, &k, &n); val = fwd_mat(g, P, 4, n, k); }          // g_h2 = [W3s W3p] [g_t; g_p]
            else if ((r -= b4) < b3) { frag_decode(r, WT, &k, &n); val = fwd_mat(g, P, 3, n, k); } // g_h1 = W2' g_a2
            else if ((r -= b3) < b2) { frag_decode(r, WT, &k, &n); val = fwd_mat(g, P, 2, n, k); } // g_v1 += W1' g_a1
            else { r -= b2; frag_decode(r, NZT, &k, &n); val = fwd_mat(g, P, 1, n, k); }           // g_x = Wa g_v
            dst[q] = (float)val;
            continue;
        }
        q -= n_bp;
        plan[g.off_winv + (size_t)blk * n_wi + q] = (float)sb[q];   // W^-1, natural layout
    }
}

size_t lsnf_prep_scratch_bytes(int nz, int depth) { return sizeof(double) * scratch_block_doubles(nz) * (size_t)depth; }

hipError_t lsnf_launch_prepare(const LsnfGeo& g, const float* const* params_host, float* plan, void* scratch,
                               hipStream_t stream) {
    LsnfParamPtrs pp;
    for (int i = 0; i < g.depth * 12; ++i) pp.p[i] = params_host[i];
    for (int i = g.depth * 12; i < LSNF_MAX_DEPTH * 12; ++i) pp.p[i] = nullptr;
    const int n = g.nz;
    const size_t lds = sizeof(double) * ((size_t)n * (n + 1) + 2 * n) + sizeof(int) * (2 * n + 4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lsnf_gj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(lsnf_gj_kernel, dim3(g.depth), dim3(256), lds, stream, pp, g.nz, (double*)scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int per_block = g.fwd_const_floats + g.fwd_block_floats + g.inv_const_floats + g.inv_block_floats +
                          g.bwd_block_floats + g.nz * g.nz;
    int gx = (per_block + 255) / 256;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(lsnf_pack_kernel, dim3(gx, g.depth), dim3(256), 0, stream, pp, g, (const double*)scratch, plan);
    return hipGetLastError();
}
