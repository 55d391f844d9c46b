// lsnf_layout.h -- geometry and prepared-weight ("plan") layout shared by host and device code.
//
// Register/lane convention used by every kernel ("sample-on-lane"):
//   a wave owns 32 samples; lane l = (m = l & 31 -> sample, h = l >> 5 -> feature half-group).
//   A 32-feature tile of activations lives in 16 VGPRs per lane: register r of lane (m,h) holds
//   feature  o(r,h) = (r & 3) + 8*(r >> 2) + 4*h  of sample m.  This is exactly the C/D layout
//   of v_mfma_f32_32x32x2_f32 (row = feature, col = sample), and -- because the k index of the
//   MFMA's B operand is (lane >> 5) -- also exactly the B-operand layout of the next GEMM when
//   register r is used as k-step r.  So activations flow GEMM -> GEMM without leaving VGPRs.
//
// "Split-pad" feature coordinates of a latent row (nz features, half = nz/2):
//   tiles 0..HT-1 hold the first half (features 0..half-1, zero padded to 32*HT),
//   tiles HT..2HT-1 hold the second half.   c -> natural column: nat(c).
//
// A weight "panel" = the 32 output features of one n-tile for all KT k-tiles of a GEMM stage,
// stored in fragment order so that one ds_read_b128 per lane feeds 4 MFMAs:
//   panel[((kt*4 + g)*64 + lane)*4 + j] = M[k = 32*kt + 8*g + 4*(lane>>5) + j][n = 32*nt + (lane&31)]
#pragma once
#include <stddef.h>

#define LSNF_MAX_DEPTH 16
#define LSNF_TILE 32
#define LSNF_FRAG_FLOATS 1024 /* one (nt,kt) 32x32 fragment block */
// Split-bf16 forward stream (lsnf_fwd3.hip): the same (nt,kt) block as THREE bf16 matrices w = w1 + w2 + w3 (each the
// round-to-nearest bf16 of what the previous ones left), in the A-operand order of v_mfma_f32_32x32x16_bf16:
//   dword[((s*3 + part)*64 + lane)*4 + jw] = pack(bf16 part of M[k(2jw)][n], M[k(2jw+1)][n]),  n = 32*nt + (lane&31),
//   k(j) = 32*kt + 16*s + (j&3) + 8*(j>>2) + 4*(lane>>5)      (s = k-step 0/1, j = 0..7)
// i.e. k-slot j of lane-half h in k-step s is accumulator register 8*s + j of that lane-half -- an output tile
// converted to bf16 pairs in register order is directly the next GEMM's B operand.
#define LSNF_FRAG3_FLOATS 1536 /* 2 k-steps x 3 parts x 1 KiB, in 4-byte units */
#define LSNF_GUARD_WORDS 256 /* word 0: folded weights outside fp16's range (written by lsnf_prepare only); rest reserved */
// fp16 range guard (lsnf_fwd2h.hip / lsnf_rev2h.hip): a wave whose GEMMs met an operand outside fp16's range writes this
// quiet-NaN bit pattern into the FIRST output element it owns (forward: logdet_out[first row of the wave]; reverse:
// z_out[first row][0]) instead of the value; the bf16x3 fix-up pass queued behind the launch recomputes every workgroup
// that finds the pattern in one of its waves' elements.  The flag lives in the launch's own output: launches in flight
// on any number of streams (or replayed from any number of graphs) cannot see each other's.
#define LSNF_F16_SENTINEL_BITS 0x7FD0F16Au
#define LSNF_F16_GUARD_MAX 65504.0f /* |x| >= this does not survive the round-to-nearest fp16 conversion */
#define LSNF_FRAG2H_FLOATS 1024 /* fp16 two-term split: 2 feature halves x 2 parts x 1 KiB */
// The same three bf16 matrices once more in the A-operand order of v_mfma_f32_16x16x32_bf16 (lsnf_fwd3.hip, 16x16 variant;
// that shape sustains a higher clock on real data).  Lane layout of that variant: a wave's 32 samples are two sample
// tiles st of 16, lane = (n = lane & 15 -> sample 16*st + n, g = lane >> 4); a 32-feature activation tile is 16 registers
// per lane, register (2*ft + st)*4 + r holding feature 16*ft + 4*g + r of sample 16*st + n.  Per (nt,kt) block:
//   dword[((ft*3 + part)*64 + lane)*4 + jw] = pack(bf16 part of M[k(2jw)][n], M[k(2jw+1)][n]),  n = 32*nt + 16*ft + (lane&15),
//   k(j) = 32*kt + (j < 4 ? 4*(lane>>4) + j : 16 + 4*(lane>>4) + j - 4)       (j = 0..7)
// i.e. k-slots 0..3 / 4..7 of lane group g are registers r = 0..3 of the ft = 0 / 1 accumulators of that lane.

struct LsnfGeo {
    int nz, half, width, depth, coupling;
    int HT, WT, NZT;            // tiles of 32: half, width, 2*HT
    // forward stream, per block
    int fwd_panels;             // panels per block
    int fwd_block_floats;       // panel floats per block
    int fwd_const_floats;       // bias/const floats per block
    // inverse stream, per block
    int inv_panels, inv_block_floats, inv_const_floats;
    // backward-z stream, per block
    int bwd_panels, bwd_block_floats, bwd_const_floats;
    // offsets (floats) into the plan buffer
    size_t off_fwd_const, off_fwd_panels;
    size_t off_inv_const, off_inv_panels;
    size_t off_bwd_const, off_bwd_panels;
    size_t off_winv;            // depth * nz*nz fp32 W^-1 (natural layout; used by d log|det W|/dW)
    int f3_block_floats;        // split-bf16 forward panels per block (same tile order as the forward stream)
    size_t off_f3_panels;
    size_t off_f3b_panels;      // the same in the 16x16x32 operand order (same size)
    int b3_block_floats;        // backward-z panels (B4, B3, B2, B1) as three bf16 matrices, 16x16x32 operand order
    size_t off_b3b_panels;
    int i3_block_floats;        // inverse panel I1 likewise
    size_t off_i3b_panels;
    int f2h_block_floats;       // forward panels as two fp16 matrices (LSNF_MATH_FP16X2), 16x16x32 operand order
    size_t off_f2h_panels;
    int i2h_block_floats;       // inverse panel I1 likewise (lsnf_rev2h.hip)
    size_t off_i2h_panels;
    size_t off_guard;           // LSNF_GUARD_WORDS 32-bit words; [0] = 1 if a folded weight is outside fp16's range (set by
                                // lsnf_prepare, read-only afterwards): the fp16 kernels then leave every row to the fix-up pass
    size_t total_floats;
};

// optional tail of the backward kernel: the Langevin update of train.py:324-329
// in-kernel noise of the Langevin update: counter-based (Philox4x32-10), a pure function of
// (seed, offset, global row, column) -- see lsnf_device.h lsnf_noise_tile and oracle/philox_oracle.py
struct LsnfRngArgs {
    unsigned long long seed, offset;
    const unsigned long long* offset_dev;   // NULL, or a device counter added to `offset` (for captured graphs)
    long long row0;                         // global index of this call's first row (sharded chains)
    int enabled;
};
struct LsnfLangevinArgs {
    const float* z_cur; const float* grad_g; const float* noise;
    float* z_new; float* gf_norm; float* gg_norm; float step;
    LsnfRngArgs rng;
};

static inline int lsnf_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Chooses the kernel instantiation (HT, WT) in {(1,1),(2,2),(2,4)} that covers (half, width).
static inline int lsnf_pick_tiles(int half, int width, int* HT, int* WT) {
    int ht = lsnf_ceil_div(half, LSNF_TILE), wt = lsnf_ceil_div(width, LSNF_TILE);
    if (ht <= 1 && wt <= 1) { *HT = 1; *WT = 1; return 0; }
    if (ht <= 2 && wt <= 2) { *HT = 2; *WT = 2; return 0; }
    if (ht <= 2 && wt <= 4) { *HT = 2; *WT = 4; return 0; }
    return -1;
}

// Additive coupling (coupling = 0, reference model.py:407-408: z2 = z2 + f(z1), f has nz/2 outputs) runs on the same
// kernels: its pre-sigmoid panels are packed as zeros with bias +40, so that sigmoid = 1 and log sigmoid = 0 exactly
// (the mechanism that already neutralises padded lanes); only the packing and un-folding differ.
// Forward stages per block (affine coupling):
//   S1  v  = Wa^T x + ca      K = NZT tiles, N = NZT tiles   (actnorm folded into the 1x1 "conv")
//   S2  h1 = relu(W1'^T v1 + c1)   K = HT, N = WT
//   S3  h2 = relu(W2'^T h1 + c2)   K = WT, N = WT
//   S4  t  = W3s^T h2 + c3s        K = WT, N = HT ;  p = W3p^T h2 + c3p   K = WT, N = HT
// Inverse stages per block:
//   I2,I3,I4 = S2,S3,S4 (same matrices) ; I1  z = Winv'^T [v1; v2] + cinv   K = NZT, N = NZT
// Backward-z stages per block (given g on [v1 | y2], recompute of S2..S4 uses the forward stream):
//   B4  g_h2 = W3s g_t + W3p g_p    K = 2*HT, N = WT
//   B3  g_h1 = W2' g_a2             K = WT,   N = WT
//   B2  g_v1 += W1' g_a1            K = WT,   N = HT
//   B1  g_x  = Wa [g_v1; g_v2]      K = NZT,  N = NZT
static inline int lsnf_geo_init(LsnfGeo* g, int nz, int width, int depth, int coupling) {
    if (nz < 2 || (nz & 1) || nz > 128 || width < 1 || width > 128 || depth < 1 || depth > LSNF_MAX_DEPTH) return -1;
    if (coupling != 0 && coupling != 1) return -1;
    g->nz = nz; g->half = nz / 2; g->width = width; g->depth = depth; g->coupling = coupling;
    if (lsnf_pick_tiles(g->half, width, &g->HT, &g->WT)) return -1;
    g->NZT = 2 * g->HT;
    const int HT = g->HT, WT = g->WT, NZT = g->NZT;
    g->fwd_panels = NZT + WT + WT + 2 * HT;
    g->fwd_block_floats = LSNF_FRAG_FLOATS * (NZT * NZT + WT * HT + WT * WT + 2 * HT * WT);
    g->fwd_const_floats = 32 * g->fwd_panels + 32;
    g->inv_panels = NZT;                       // I1 only; I2..I4 reuse the forward panels
    g->inv_block_floats = LSNF_FRAG_FLOATS * (NZT * NZT);
    g->inv_const_floats = 32 * NZT;
    g->bwd_panels = WT + WT + HT + NZT;
    g->bwd_block_floats = LSNF_FRAG_FLOATS * (WT * 2 * HT + WT * WT + HT * WT + NZT * NZT);
    g->bwd_const_floats = 0;
    size_t o = 0;
    g->off_fwd_const = o;  o += (size_t)depth * g->fwd_const_floats;
    o = (o + 255) & ~(size_t)255;
    g->off_fwd_panels = o; o += (size_t)depth * g->fwd_block_floats;
    g->off_inv_const = o;  o += (size_t)depth * g->inv_const_floats;
    o = (o + 255) & ~(size_t)255;
    g->off_inv_panels = o; o += (size_t)depth * g->inv_block_floats;
    g->off_bwd_const = o;
    g->off_bwd_panels = o; o += (size_t)depth * g->bwd_block_floats;
    g->off_winv = o;       o += (size_t)depth * nz * nz;
    o = (o + 255) & ~(size_t)255;
    g->f3_block_floats = LSNF_FRAG3_FLOATS * (NZT * NZT + WT * HT + WT * WT + 2 * HT * WT);
    g->off_f3_panels = o;  o += (size_t)depth * g->f3_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->off_f3b_panels = o; o += (size_t)depth * g->f3_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->b3_block_floats = LSNF_FRAG3_FLOATS * (WT * 2 * HT + WT * WT + HT * WT + NZT * NZT);
    g->off_b3b_panels = o; o += (size_t)depth * g->b3_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->i3_block_floats = LSNF_FRAG3_FLOATS * (NZT * NZT);
    g->off_i3b_panels = o; o += (size_t)depth * g->i3_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->f2h_block_floats = LSNF_FRAG2H_FLOATS * (NZT * NZT + WT * HT + WT * WT + 2 * HT * WT);
    g->off_f2h_panels = o; o += (size_t)depth * g->f2h_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->i2h_block_floats = LSNF_FRAG2H_FLOATS * (NZT * NZT);
    g->off_i2h_panels = o; o += (size_t)depth * g->i2h_block_floats;
    o = (o + 255) & ~(size_t)255;
    g->off_guard = o; o += 256;
    g->total_floats = o;
    return 0;
}

// ---- workspace of the parameter-gradient path (lsnf_backward_params) -----------------------------
// [0,4)                 : G = sum_b dL/dlogdet_b (+3 pad)
// folded gradients      : depth * LsnfFoldLayout.per_block floats (zeroed every call, accumulated by atomics)
// dump                  : depth * LsnfDumpLayout.per_block floats (per-sample intermediates of the backward)
#define LSNF_AL4(x) (((x) + 3) & ~(size_t)3)
struct LsnfDumpLayout { size_t off_gv, off_ga1, off_ga2, off_gt, off_gp, off_h1, off_h2, per_block; };
#ifdef __HIPCC__
__host__ __device__
#endif
static inline LsnfDumpLayout lsnf_dump_layout(int B, int nz, int width) {
    // (rows rounded up to whole 32-sample tiles: the tiled form of the large-batch path writes whole tiles, lsnf_l16.h l16_store_tiled)
    LsnfDumpLayout d; size_t o = 0; const size_t b = ((size_t)B + 31) / 32 * 32; const int half = nz / 2;
    d.off_gv = o;  o = LSNF_AL4(o + b * nz);
    d.off_ga1 = o; o = LSNF_AL4(o + b * width);
    d.off_ga2 = o; o = LSNF_AL4(o + b * width);
    d.off_gt = o;  o = LSNF_AL4(o + b * half);
    d.off_gp = o;  o = LSNF_AL4(o + b * half);
    d.off_h1 = o;  o = LSNF_AL4(o + b * width);
    d.off_h2 = o;  o = LSNF_AL4(o + b * width);
    d.per_block = o;
    return d;
}
struct LsnfFoldLayout { int dWa, dca, dW1, dc1, dW2, dc2, dW3s, dc3s, dW3p, dc3p, per_block; };
#ifdef __HIPCC__
__host__ __device__
#endif
static inline LsnfFoldLayout lsnf_fold_layout(int nz, int width) {
    LsnfFoldLayout f; int o = 0; const int half = nz / 2, w = width;
    f.dWa = o; o += nz * nz;   f.dca = o; o += nz;
    f.dW1 = o; o += half * w;  f.dc1 = o; o += w;
    f.dW2 = o; o += w * w;     f.dc2 = o; o += w;
    f.dW3s = o; o += w * half; f.dc3s = o; o += half;
    f.dW3p = o; o += w * half; f.dc3p = o; o += half;
    f.per_block = (o + 3) & ~3;
    return f;
}
// ... + 4 floats behind the dump: word 0 = the layout of the h1 / h2 arrays the forward of this evaluation wrote (0 row-major,
// 1 tiled), set by lsnf_forward for the batch sizes at which the contraction of lsnf_params3.hip may run
static inline size_t lsnf_params_workspace_tag(int nz, int width, int depth, int B) {
    return 4 + (size_t)depth * lsnf_fold_layout(nz, width).per_block + (size_t)depth * lsnf_dump_layout(B, nz, width).per_block;
}
static inline size_t lsnf_params_workspace_floats(int nz, int width, int depth, int B) {
    return lsnf_params_workspace_tag(nz, width, depth, B) + 4;
}
// the tiled form of the dump needs whole feature groups of 16 and the natural tile order of the latent rows
static inline int lsnf_dump_can_tile(int nz, int width) { return nz % 64 == 0 && width % 16 == 0; }
#define LSNF_X3_MIN_ROWS 12288      /* lsnf_params3.hip takes the batch contraction from this many rows (fp32-MFMA kernel below) */

// ---- optional activation stash of the forward (act_saved), read by the backward instead of recomputing the MLP ----
// Opaque, register-order layout, per block and per 32-sample tile `wt` (nwt = ceil(B/32) tiles):
//   floats [0, HT*1024)          : sigma tiles, each [g(4)][lane(64)][4]  (sigma of feature tile t, accumulator layout)
//   words  [HT*1024, +2*WT*64)   : relu masks, uint32 per lane: tiles 0..WT-1 of h1, then 0..WT-1 of h2 (bit r = h[r] > 0)
struct LsnfActLayout { size_t per_tile, per_block, mask_off; };
#ifdef __HIPCC__
__host__ __device__
#endif
static inline LsnfActLayout lsnf_act_layout(int B, int HT, int WT) {
    LsnfActLayout a;
    a.mask_off = (size_t)HT * 1024;
    a.per_tile = a.mask_off + (size_t)2 * WT * 64;
    a.per_block = a.per_tile * (size_t)((B + 31) / 32);
    return a;
}
