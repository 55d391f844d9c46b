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
// (1) Gauss-Jordan with partial pivoting, float64, one workgroup (T x T threads) per block.
// The matrix lives in REGISTERS: thread (ry, cx) owns the E x E elements A[ry + T*rr][cx + T*jj], E = 128 / T.
// Per column only the pivot column, the pivot row and the displaced row travel through LDS
// (ping-pong buffers -> 2 barriers per column); every wave finds the pivot redundantly by shuffles.
// T = 32 (1 024 threads, 4 x 4 elements each): a column step is a chain of dependent instructions (publish, barrier, pivot
// search, fp64 reciprocal, barrier, eliminate), so four waves per SIMD with a quarter of the per-thread work each run it
// ~3x faster than T = 16 with one wave per SIMD (tools/prep_loop.py under rocprofv3: 241 us -> see profiles/).
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T * T) void lsnf_gj_kernel(LsnfParamPtrs pp, int nz, double* scratch) {
    constexpr int E = 128 / T;
    __shared__ double colbuf[2][128], rowp[2][128], rowc[2][128];
    __shared__ double pivs[128];
    __shared__ int perm[128], cinv[128], cmap[128];
    const int n = nz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ry = tid / T, cx = tid % T;
    const int blk = blockIdx.x;
    const float* W = pp.p[blk * 12 + P_W];
    double a[E][E];
#pragma unroll
    for (int rr = 0; rr < E; ++rr)
#pragma unroll
        for (int jj = 0; jj < E; ++jj) {
            const int r = ry + T * rr, j = cx + T * jj;
            a[rr][jj] = (r < n && j < n) ? (double)W[r * n + j] : ((r == j) ? 1.0 : 0.0);   // identity padding
        }
    for (int c = 0; c < n; ++c) {
        const int pb = c & 1, cjj = c / T, ccx = c % T;
        // (A) owners of column c publish it
        if (cx == ccx) {
#pragma unroll
            for (int jj = 0; jj < E; ++jj)
                if (jj == cjj) {
#pragma unroll
                    for (int rr = 0; rr < E; ++rr) colbuf[pb][ry + T * rr] = a[rr][jj];
                }
        }
        __syncthreads();
        // (B) pivot row: arg max over r >= c of |A[r][c]| taken at float precision (any near-maximal pivot
        //     is as good), packed with the row index into ONE 32-bit key so the wave reduction is a single
        //     shuffle per round; every wave computes it redundantly (no broadcast barrier).
        unsigned key = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = lane + 64 * q;
            if (r >= c && r < n) {
                const unsigned k = (__float_as_uint((float)fabs(colbuf[pb][r])) & 0xFFFFFF80u) | (unsigned)(127 - r);
                key = k > key ? k : key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned ok = __shfl_xor(key, o, 64); key = ok > key ? ok : key; }
        const int p = 127 - (int)(key & 127u);
        const double pv = colbuf[pb][p];
        const double old_cc = colbuf[pb][c];          // A[c][c] before the exchange (multiplier of row p afterwards)
        const int prr = p / T, pry = p % T, crr = c / T, cry = c % T;
        // (C) owners of rows p and c publish them
        if (ry == pry) {
#pragma unroll
            for (int rr = 0; rr < E; ++rr)
                if (rr == prr) {
#pragma unroll
                    for (int jj = 0; jj < E; ++jj) rowp[pb][cx + T * jj] = a[rr][jj];
                }
        }
        if (ry == cry) {
#pragma unroll
            for (int rr = 0; rr < E; ++rr)
                if (rr == crr) {
#pragma unroll
                    for (int jj = 0; jj < E; ++jj) rowc[pb][cx + T * jj] = a[rr][jj];
                }
        }
        if (tid == 0) perm[c] = p;
        __syncthreads();
        // (D) exchange rows c <-> p, scale the pivot row (pivot slot -> 1/pv), eliminate column c elsewhere.
        //     Generic rows: a -= f * pr with column c zeroed first (in-place inverse); rows c and p are then
        //     overwritten by their owners -- keeps the 64-FMA body free of per-element selects.
        const double ipv = 1.0 / pv;
        if (tid == 0) pivs[c] = pv;
        double pr[E], f[E];
#pragma unroll
        for (int jj = 0; jj < E; ++jj) { const int j = cx + T * jj; pr[jj] = ((j == c) ? 1.0 : rowp[pb][j]) * ipv; }
#pragma unroll
        for (int rr = 0; rr < E; ++rr) f[rr] = colbuf[pb][ry + T * rr];
        if (cx == ccx) {
#pragma unroll
            for (int jj = 0; jj < E; ++jj)
                if (jj == cjj) {
#pragma unroll
                    for (int rr = 0; rr < E; ++rr) a[rr][jj] = 0.0;
                }
        }
#pragma unroll
        for (int rr = 0; rr < E; ++rr)
#pragma unroll
            for (int jj = 0; jj < E; ++jj) a[rr][jj] = fma(-f[rr], pr[jj], a[rr][jj]);
        if (ry == pry && p != c) {          // row p now holds the old row c, eliminated with multiplier old A[c][c]
#pragma unroll
            for (int rr = 0; rr < E; ++rr)
                if (rr == prr) {
#pragma unroll
                    for (int jj = 0; jj < E; ++jj) {
                        const int j = cx + T * jj;
                        a[rr][jj] = fma(-old_cc, pr[jj], (j == c) ? 0.0 : rowc[pb][j]);
                    }
                }
        }
        if (ry == cry) {                    // row c becomes the scaled pivot row
#pragma unroll
            for (int rr = 0; rr < E; ++rr)
                if (rr == crr) {
#pragma unroll
                    for (int jj = 0; jj < E; ++jj) a[rr][jj] = pr[jj];
                }
        }
    }
    // undo the row interchanges as column interchanges: out[:, cinv[j]] = A[:, j]
    __syncthreads();
    if (tid == 0) {
        for (int j = 0; j < 128; ++j) cmap[j] = j;
        for (int c = n - 1; c >= 0; --c) { const int p = perm[c]; if (p != c) { const int t = cmap[c]; cmap[c] = cmap[p]; cmap[p] = t; } }
        for (int j = 0; j < n; ++j) cinv[cmap[j]] = j;   // position j of the result holds work-column cmap[j]
    }
    __syncthreads();
    double* out = scratch + (size_t)blk * scratch_block_doubles(nz);
#pragma unroll
    for (int rr = 0; rr < E; ++rr)
#pragma unroll
        for (int jj = 0; jj < E; ++jj) {
            const int r = ry + T * rr, j = cx + T * jj;
            if (r < n && j < n) out[r * n + cinv[j]] = a[rr][jj];
        }
    {   // log|det W| = sum log|pivot| and sum(3 logs): one term per thread, summed by wave 0 and wave 1
        const float* logs = pp.p[blk * 12 + P_ALOGS];
        // reference: torch.sum(logs * 3) in fp32 (model.py:264,273); each term logs*3 is rounded
        // to fp32 first, then summed -- we sum those fp32 terms in double and round once.
        if (tid < 128) {
            double la = tid < n ? log(fabs(pivs[tid])) : 0.0, s3 = tid < n ? (double)(logs[tid] * 3.0f) : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { la += __shfl_xor(la, o, 64); s3 += __shfl_xor(s3, o, 64); }
            if (lane == 0) { colbuf[0][tid >> 6] = la; rowp[0][tid >> 6] = s3; }
        }
        __syncthreads();
        if (tid == 0) {
            out[(size_t)n * n + 0] = colbuf[0][0] + colbuf[0][1];
            out[(size_t)n * n + 1] = rowp[0][0] + rowp[0][1];
        }
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
        if (g.coupling == 0) {   // additive: fc_zeros is (w, nz/2), all shift; no scale path
            if (which) return 0.0;
            return (double)P[P_W3][k * half + f] * e3(P[P_LOGS3], f);
        }
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
        if (g.coupling == 0) return which ? 40.0 : (double)P[P_B3][f] * e3(P[P_LOGS3], f);
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

// split-bf16 stream: dword index inside a stage's panels -> the two (k, n) it packs and which part (0..2)
__device__ static inline void frag3_decode(int q, int KT, int* k0, int* k1, int* n, int* part) {
    const int per_panel = KT * LSNF_FRAG3_FLOATS;
    const int nt = q / per_panel; int r = q % per_panel;
    const int kt = r / LSNF_FRAG3_FLOATS; r %= LSNF_FRAG3_FLOATS;
    const int sp = r / 256; r %= 256;                 // s*3 + part
    const int s = sp / 3, lane = r / 4, jw = r % 4;
    *part = sp % 3;
    const int j0 = 2 * jw, j1 = 2 * jw + 1, h = lane >> 5;
    *k0 = 32 * kt + 16 * s + (j0 & 3) + 8 * (j0 >> 2) + 4 * h;
    *k1 = 32 * kt + 16 * s + (j1 & 3) + 8 * (j1 >> 2) + 4 * h;
    *n = 32 * nt + (lane & 31);
}
// the same for the 16x16x32 operand order (lsnf_layout.h)
__device__ static inline void frag3b_decode(int q, int KT, int* k0, int* k1, int* n, int* part) {
    const int per_panel = KT * LSNF_FRAG3_FLOATS;
    const int nt = q / per_panel; int r = q % per_panel;
    const int kt = r / LSNF_FRAG3_FLOATS; r %= LSNF_FRAG3_FLOATS;
    const int fp = r / 256; r %= 256;                 // ft*3 + part
    const int ft = fp / 3, lane = r / 4, jw = r % 4;
    *part = fp % 3;
    const int j0 = 2 * jw, j1 = 2 * jw + 1, kg = lane >> 4;
    *k0 = 32 * kt + (j0 < 4 ? 4 * kg + j0 : 16 + 4 * kg + j0 - 4);
    *k1 = 32 * kt + (j1 < 4 ? 4 * kg + j1 : 16 + 4 * kg + j1 - 4);
    *n = 32 * nt + 16 * ft + (lane & 15);
}
// the fp16 two-term stream (lsnf_fwd2h.hip): same operand order, parts 0..1
__device__ static inline void frag2h_decode(int q, int KT, int* k0, int* k1, int* n, int* part) {
    const int per_panel = KT * LSNF_FRAG2H_FLOATS;
    const int nt = q / per_panel; int r = q % per_panel;
    const int kt = r / LSNF_FRAG2H_FLOATS; r %= LSNF_FRAG2H_FLOATS;
    const int fp = r / 256; r %= 256;                 // ft*2 + part
    const int ft = fp / 2, lane = r / 4, jw = r % 4;
    *part = fp % 2;
    const int j0 = 2 * jw, j1 = 2 * jw + 1, kg = lane >> 4;
    *k0 = 32 * kt + (j0 < 4 ? 4 * kg + j0 : 16 + 4 * kg + j0 - 4);
    *k1 = 32 * kt + (j1 < 4 ? 4 * kg + j1 : 16 + 4 * kg + j1 - 4);
    *n = 32 * nt + 16 * ft + (lane & 15);
}
// part p (0..1) of the error-free split w = w1 + w2 into round-to-nearest fp16 terms, as its 16-bit pattern
__device__ static inline unsigned f16_part_bits(double v, int part) {
    float r = (float)v;
    _Float16 h = (_Float16)r;
    if (part == 1) h = (_Float16)(r - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
// round-to-nearest-even bf16 of a float, as its 16-bit pattern (finite inputs)
__device__ static inline unsigned bf16_rne_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// part p (0..2) of the error-free split w = w1 + w2 + w3 of the fp32 value the fp32 stream holds
__device__ static inline unsigned bf16_part_bits(double v, int part) {
    float r = (float)v;
    unsigned b = bf16_rne_bits(r);
    for (int i = 0; i < part; ++i) { r = r - __uint_as_float(b << 16); b = bf16_rne_bits(r); }
    return b & 0xFFFFu;
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
    const int n_bp = g.bwd_block_floats, n_wi = nz * nz, n_f3 = g.f3_block_floats;
    const int n_b3 = g.b3_block_floats, n_i3 = g.i3_block_floats;
    const int total = n_fc + n_fp + n_ic + n_ip + n_bp + n_wi + 2 * n_f3 + n_b3 + n_i3;
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
        if (q < n_wi) { plan[g.off_winv + (size_t)blk * n_wi + q] = (float)sb[q]; continue; }   // W^-1, natural layout
        q -= n_wi;
        if (q >= 2 * n_f3 + n_b3) {         // ---- split-bf16 inverse panel I1: Winv' = Winv diag(exp(-3 logs_a)), 16x16x32 order
            q -= 2 * n_f3 + n_b3;
            unsigned* dst = reinterpret_cast<unsigned*>(plan + g.off_i3b_panels + (size_t)blk * n_i3);
            int k0, k1, n, part;
            frag3b_decode(q, NZT, &k0, &k1, &n, &part);
            const SplitIdx sn = split_nat(n, HT, g.half);
            double v[2] = {0.0, 0.0};
            const int ks[2] = {k0, k1};
            for (int i = 0; i < 2; ++i) {
                const SplitIdx sk = split_nat(ks[i], HT, g.half);
                if (sk.ok && sn.ok) v[i] = sb[(size_t)sk.nat * nz + sn.nat] * exp(-(double)(P[P_ALOGS][sn.nat] * 3.0f));
            }
            dst[q] = bf16_part_bits(v[0], part) | (bf16_part_bits(v[1], part) << 16);
            continue;
        }
        if (q >= 2 * n_f3) {                // ---- split-bf16 backward panels (transposes, as the fp32 backward stream), 16x16x32 order
            q -= 2 * n_f3;
            unsigned* dst = reinterpret_cast<unsigned*>(plan + g.off_b3b_panels + (size_t)blk * n_b3);
            int r = q, k0, k1, n, part;
            const int b4 = LSNF_FRAG3_FLOATS * WT * 2 * HT, b3 = LSNF_FRAG3_FLOATS * WT * WT, b2 = LSNF_FRAG3_FLOATS * HT * WT;
            int stage, KT;
            if (r < b4) { stage = 4; KT = 2 * HT; }
            else if ((r -= b4) < b3) { stage = 3; KT = WT; }
            else if ((r -= b3) < b2) { stage = 2; KT = WT; }
            else { r -= b2; stage = 1; KT = NZT; }
            frag3b_decode(r, KT, &k0, &k1, &n, &part);
            dst[q] = bf16_part_bits(fwd_mat(g, P, stage, n, k0), part) | (bf16_part_bits(fwd_mat(g, P, stage, n, k1), part) << 16);
            continue;
        }
        {                                   // ---- split-bf16 forward panels (same matrices as the forward stream), both operand orders
            const bool shape16 = q >= n_f3;
            if (shape16) q -= n_f3;
            unsigned* dst = reinterpret_cast<unsigned*>(plan + (shape16 ? g.off_f3b_panels : g.off_f3_panels) + (size_t)blk * n_f3);
            int r = q, k0, k1, n, part, stage, KT;
            const int s1 = LSNF_FRAG3_FLOATS * NZT * NZT, s2 = LSNF_FRAG3_FLOATS * WT * HT, s3 = LSNF_FRAG3_FLOATS * WT * WT;
            if (r < s1) { stage = 1; KT = NZT; }
            else if ((r -= s1) < s2) { stage = 2; KT = HT; }
            else if ((r -= s2) < s3) { stage = 3; KT = WT; }
            else { r -= s3; stage = 4; KT = WT; }
            if (shape16) frag3b_decode(r, KT, &k0, &k1, &n, &part); else frag3_decode(r, KT, &k0, &k1, &n, &part);
            dst[q] = bf16_part_bits(fwd_mat(g, P, stage, k0, n), part) | (bf16_part_bits(fwd_mat(g, P, stage, k1, n), part) << 16);
        }
    }
    // ---- fp16 two-term forward panels (LSNF_MATH_FP16X2), 16x16x32 operand order
    for (int q = blockIdx.x * 256 + threadIdx.x; q < g.f2h_block_floats; q += gridDim.x * 256) {
        unsigned* dst = reinterpret_cast<unsigned*>(plan + g.off_f2h_panels + (size_t)blk * g.f2h_block_floats);
        int r = q, k0, k1, n, part, stage, KT;
        const int s1 = LSNF_FRAG2H_FLOATS * NZT * NZT, s2 = LSNF_FRAG2H_FLOATS * WT * HT, s3 = LSNF_FRAG2H_FLOATS * WT * WT;
        if (r < s1) { stage = 1; KT = NZT; }
        else if ((r -= s1) < s2) { stage = 2; KT = HT; }
        else if ((r -= s2) < s3) { stage = 3; KT = WT; }
        else { r -= s3; stage = 4; KT = WT; }
        frag2h_decode(r, KT, &k0, &k1, &n, &part);
        const double w0 = fwd_mat(g, P, stage, k0, n), w1 = fwd_mat(g, P, stage, k1, n);
        if (!(fabs(w0) < (double)LSNF_F16_GUARD_MAX) || !(fabs(w1) < (double)LSNF_F16_GUARD_MAX))     // also NaN
            reinterpret_cast<unsigned*>(plan + g.off_guard)[0] = 1u;                                  // fp16 forward unusable: lsnf_fwd2h.hip defers to bf16x3
        dst[q] = f16_part_bits(w0, part) | (f16_part_bits(w1, part) << 16);
    }
    // ---- fp16 two-term inverse panel I1 (lsnf_rev2h.hip): Winv' = Winv diag(exp(-3 logs_a)), 16x16x32 operand order
    for (int q = blockIdx.x * 256 + threadIdx.x; q < g.i2h_block_floats; q += gridDim.x * 256) {
        unsigned* dst = reinterpret_cast<unsigned*>(plan + g.off_i2h_panels + (size_t)blk * g.i2h_block_floats);
        int k0, k1, n, part;
        frag2h_decode(q, NZT, &k0, &k1, &n, &part);
        const SplitIdx sn = split_nat(n, HT, g.half);
        double v[2] = {0.0, 0.0};
        const int ks[2] = {k0, k1};
        for (int i = 0; i < 2; ++i) {
            const SplitIdx sk = split_nat(ks[i], HT, g.half);
            if (sk.ok && sn.ok) v[i] = sb[(size_t)sk.nat * nz + sn.nat] * exp(-(double)(P[P_ALOGS][sn.nat] * 3.0f));
        }
        if (!(fabs(v[0]) < (double)LSNF_F16_GUARD_MAX) || !(fabs(v[1]) < (double)LSNF_F16_GUARD_MAX))
            reinterpret_cast<unsigned*>(plan + g.off_guard)[0] = 1u;
        dst[q] = f16_part_bits(v[0], part) | (f16_part_bits(v[1], part) << 16);
    }
}

size_t lsnf_prep_scratch_bytes(int nz, int depth) { return sizeof(double) * scratch_block_doubles(nz) * (size_t)depth; }

hipError_t lsnf_launch_prepare(const LsnfGeo& g, const float* const* params_host, float* plan, void* scratch,
                               hipStream_t stream) {
    LsnfParamPtrs pp;
    for (int i = 0; i < g.depth * 12; ++i) pp.p[i] = params_host[i];
    for (int i = g.depth * 12; i < LSNF_MAX_DEPTH * 12; ++i) pp.p[i] = nullptr;
    hipLaunchKernelGGL(lsnf_gj_kernel<32>, dim3(g.depth), dim3(1024), 0, stream, pp, g.nz, (double*)scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int per_block = g.fwd_const_floats + g.fwd_block_floats + g.inv_const_floats + g.inv_block_floats +
                          g.bwd_block_floats + g.nz * g.nz + 2 * g.f3_block_floats + g.b3_block_floats + g.i3_block_floats;
    if (hipError_t em = hipMemsetAsync(plan + g.off_guard, 0, sizeof(unsigned) * LSNF_GUARD_WORDS, stream); em != hipSuccess) return em;
    int gx = (per_block + 255) / 256;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(lsnf_pack_kernel, dim3(gx, g.depth), dim3(256), 0, stream, pp, g, (const double*)scratch, plan);
    return hipGetLastError();
}
