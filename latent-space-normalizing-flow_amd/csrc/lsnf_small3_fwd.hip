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
// (192 KiB per block) through the CU's vector-memory path.  Weights go L2 -> VGPR one block ahead, each fragment register
// re-loaded right after its last use, one load behind every MFMA (lsnf_small3.h units_mma_st), never under a branch.
//
// Lane layout "L16" (lsnf_layout.h): lane = (n = lane & 15 -> sample, g = lane >> 4); a half-unit (16 features x 16
// samples) is 4 registers per lane: feature 16*ft + 4*g + r.
#include <stdlib.h>
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
};
// LDS map (floats) of a workgroup that owns ST sample tiles of 16 rows: X (ST x NZT B-tiles; ONE buffer: with the coupling of
// block b deferred into block b+1's first S1 half, every tile is rewritten only after a barrier behind its last reader), H1, H2
// (ST x WT B-tiles each), reductions (ST x 4 waves x 16 x 2), then the constant blocks
template <class C, int ST>
struct Small3Lds {
    static constexpr int XT = C::NZT * S3_BTILE_FLOATS;        // one sample tile's block input
    static constexpr int HL = C::WT * S3_BTILE_FLOATS;         // one sample tile's hidden layer
    static constexpr int L_X = 0;
    static constexpr int L_H1 = L_X + ST * XT;
    static constexpr int L_H2 = L_H1 + ST * HL;
    static constexpr int L_RED = L_H2 + ST * HL;
    static constexpr int L_CONST = L_RED + ST * 4 * 16 * 2;
};

struct Small3Args {
    const float* consts; const float* panels3b;
    const float* z_in; const float* objective;
    float* z_out; float* logdet_out; float* ll_out; float* z_saved; float* act_saved;
    double* stats;
    float* hdump;                      // NULL, or the parameter-gradient dump (LsnfDumpLayout) of block first_block: h1, h2 are written
    int B, nz, half, n_blocks, vec4, width;
    unsigned long long* stamps;        // LSNF_STAMPS diagnostic build only: [workgroup * 4 + wave, & 2047][64] clock stamps
};

#ifdef LSNF_STAMPS   // per-stage cycles of block 1 and the in-kernel clock (tools/stamps_small3.py)
#define S3_STAMP(i, INSN)                                                                               \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile(INSN " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");  \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * 4 + wave) & 2047) * 64 + (i)] = t_; } while (0)
#else
#define S3_STAMP(i, INSN) do {} while (0)
#endif

// ReLU mask of one half-unit (nt, ft) of a hidden layer into the stash (lsnf_layout.h LsnfActLayout: per 32-sample tile and hidden
// tile one 32-bit word per stash lane, bit 4*(2*ft + (g >> 1)) + r = h[r] > 0): the two lane groups g >> 1 meet in one
// v_permlane32_swap, and what a wave then holds is exactly BYTE ft of the word -- stored as a byte by lanes 0..31; no LDS staging, no atomics
// (the ft = 1 wave also clears the unused upper half).
__device__ __forceinline__ void stash_relu_mask(float* act_tile, size_t mask_off, int t, int lane32, int ft, const f32x4& h, int lane) {
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) c |= (h[r] > 0.0f ? 1u : 0u) << (4 * (lane >> 5) + r);
    c = pair_or32(c);
    if (lane < 32) {
        unsigned char* w = reinterpret_cast<unsigned char*>(reinterpret_cast<unsigned*>(act_tile + mask_off) + t * 64 + lane32);
        w[ft] = (unsigned char)c;
        if (ft) *reinterpret_cast<unsigned short*>(w + 2) = (unsigned short)0;
    }
}

// ST = sample tiles (of 16 rows) per workgroup.  ST = 1 is the latency form (B = 100: 7 workgroups).  ST = 2 / 4 are the
// SHARD-SIZE forms (strong scaling of the 65 536-row evaluation over 8 / 4 GPUs leaves 8 192 / 16 384 rows per GPU): a
// workgroup of 32 / 64 rows streams the same 192 KiB of weights per block as a 16-row one -- every fetched fragment triple
// feeds 6 * ST MFMAs -- so those launches are ONE round of <= 256 workgroups instead of two / four rounds of 16-row ones
// (tools/micro/wstream.hip: one workgroup per CU streams the 960 KiB stack in ~9 us + launch; two per CU take 1.7x as long,
// the CU's L2 port is the bound), and the weight stream hides under the MFMA chains (96 * ST * 16 cycles per wave and block).
// EXTRAS = the call keeps something for a backward pass (block outputs z_saved, the activation stash act_saved, h1 / h2 for
// the parameter gradients): the plain log-prob evaluation is compiled without those stores and their branches, so that each
// stage is one scheduling region.
// (Tried: two workgroups per CU -- __launch_bounds__(256, 2), 512 32-row workgroups for 16 384 rows, so that a second wave on the
// SIMD fills the first one's bubbles.  68 B/lane of scratch at 256 registers, and the two workgroups stream the weights twice
// through the CU's one vector-memory path: 35.1 us at 16 384 rows against 31.9 for lsnf_fwd3q_kernel, 22.4 instead of 20.1 at
// 8 192.  LSNF_SMALL3_WAVES2=1 rebuilds that form.)
#ifndef LSNF_SMALL3_WAVES2
#define LSNF_SMALL3_WAVES2 0
#endif
template <class C, int ST, bool EXTRAS>
__global__ __launch_bounds__(256, (LSNF_SMALL3_WAVES2 && ST <= 2 && C::WT <= 2 && !EXTRAS) ? 2 : 1) void lsnf_small3_fwd_kernel(const Small3Args a) {
    constexpr int HT = C::HT, WT = C::WT, NZT = C::NZT, NU2 = C::NU2;
    using L = Small3Lds<C, ST>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XB = smem + L::L_X;                                           // [st][NZT] B-tiles
    float* H1B = smem + L::L_H1;                                         // [st][WT]
    float* H2B = smem + L::L_H2;
    float* RED = smem + L::L_RED;
    float* cst = smem + L::L_CONST;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;

    S3_STAMP(0, "s_memtime");
    S3_STAMP(50, "s_memrealtime");
    // this wave's half-units: hu1 of the latent halves (v1 / v2 / t / p), hw[i] of the hidden layers.  A wave without a unit of
    // its own (HT = 1: waves 2, 3) computes unit 0 again and stores the same values to the same LDS words: no branch in the stages
    const bool has1 = wave < 2 * HT;
    const int hu1 = has1 ? wave : 0, nt1 = hu1 >> 1, ft1 = hu1 & 1;
    int hw[NU2]; bool hasw[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) { hasw[i] = wave + 4 * i < 2 * WT; hw[i] = hasw[i] ? wave + 4 * i : 0; }

    long sample[ST]; bool live[ST]; long row[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        sample[st] = ((long)blockIdx.x * ST + st) * S3_SAMPLES + n;
        live[st] = sample[st] < a.B;
        row[st] = live[st] ? sample[st] : (long)a.B - 1;
    }
    // Prologue.  The CU's vector-memory path (16 cycles per KiB, shared by the four waves) is busy for ~3 000 cycles with the
    // first block's 192 KiB of weights whatever the order; what the order decides is what the waves can do meanwhile:
    //   constant blocks by LDS-DMA (no registers, no wait of their own) -> z rows -> S1's weights (96 KiB, in order of use) ->
    //   [the rows are back: split them into B-operand tiles while the weights keep arriving] -> the other stages' weights -> barrier.
    // vmcnt completes in order: waiting for the rows never waits for a weight, and it covers the constants' DMA, which is older.
    {
        // (1 KiB pieces; the constant blocks are a multiple of 16 bytes: the last piece is cut by the lane mask)
        const int nconst = a.n_blocks * C::CONST_FLOATS;
        for (int p0 = wave * 256; p0 < nconst; p0 += 1024)
            if (p0 + 4 * lane < nconst)
                __builtin_amdgcn_global_load_lds((const LSNF_AS1 void*)(a.consts + p0 + 4 * lane), (LSNF_AS3 void*)(cst + p0), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    S3_STAMP(2, "s_memtime");
    // each wave loads the two half-units of the latent row it owns: (nt1, ft1) of the first half, (HT + nt1, ft1) of the second
    f32x4 xrow[ST][2];
    if (a.vec4 == 4) {           // (one wave-uniform branch for all row loads instead of one per load)
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j) xrow[st][j] = load_row_half<HT>(j * HT + nt1, ft1, a.z_in + row[st] * (long)a.nz, a.half, g, 4);
    } else {
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j) xrow[st][j] = load_row_half<HT>(j * HT + nt1, ft1, a.z_in + row[st] * (long)a.nz, a.half, g, a.vec4);
    }
    float ell[ST];                                                        // per-wave partial of the running log-det
#pragma unroll
    for (int st = 0; st < ST; ++st) ell[st] = (wave == 0 && a.objective) ? a.objective[row[st]] : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    S3_STAMP(3, "s_memtime");
    // from here on every fragment register is re-loaded with the next block's fragment right after its last use
    // (units_mma_st): the weight stream runs one block ahead of the MFMAs
    // (in order of use: S1's two units interleaved per k-tile; the last k-tile of S4's units comes with the first S1, below)
    UFrags<NZT> w1a, w1b;
    {
        const bf16x8* pa = unit_ptr<NZT>(a.panels3b, nt1, ft1, lane);
        const bf16x8* pb = unit_ptr<NZT>(a.panels3b, HT + nt1, ft1, lane);
#pragma unroll
        for (int kt = 0; kt < NZT; ++kt)
#pragma unroll
            for (int p = 0; p < 3; ++p) { w1a.w[kt][p] = pa[(kt * 6 + p) * 64]; w1b.w[kt][p] = pb[(kt * 6 + p) * 64]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    S3_STAMP(4, "s_memtime");
    // the first-half unit goes to LDS now; the second-half unit stays in registers: it is the "v2" of a virtual block -1 whose
    // coupling is the identity (below), run under the first block's S1 like every other block's
#pragma unroll
    for (int st = 0; st < ST; ++st) store_half(XB + st * L::XT + nt1 * S3_BTILE_FLOATS, ft1, xrow[st][0], lane);
    __builtin_amdgcn_sched_barrier(0);
    UFrags<HT> w2[NU2];
    UFrags<WT> w3[NU2];
#pragma unroll
    for (int i = 0; i < NU2; ++i) w2[i] = fetch_unit<HT>(a.panels3b + C::OFF3_S2, hw[i] >> 1, hw[i] & 1, lane);
#pragma unroll
    for (int i = 0; i < NU2; ++i) w3[i] = fetch_unit<WT>(a.panels3b + C::OFF3_S3, hw[i] >> 1, hw[i] & 1, lane);
    UFrags<WT> w4t, w4p;
    {
        const bf16x8* pt = unit_ptr<WT>(a.panels3b + C::OFF3_S4, nt1, ft1, lane);
        const bf16x8* pq = unit_ptr<WT>(a.panels3b + C::OFF3_S4, HT + nt1, ft1, lane);
#pragma unroll
        for (int kt = 0; kt < WT; ++kt)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (kt < WT - 1) { w4t.w[kt][p] = pt[(kt * 6 + p) * 64]; w4p.w[kt][p] = pq[(kt * 6 + p) * 64]; }
                else { w4t.w[kt][p] = bf16x8{}; w4p.w[kt][p] = bf16x8{}; }
            }
    }
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const bool w4 = (a.width & 3) == 0;
    // stash addressing: 16-row tile q = blockIdx.x * ST + st is half (q & 1) of the 32-sample stash tile q >> 1
    size_t wtile[ST]; int lane32[ST]; bool tile_ok[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        const size_t q = (size_t)blockIdx.x * ST + st;
        wtile[st] = q >> 1;
        tile_ok[st] = (long)q * S3_SAMPLES < (long)a.B;                   // (a 16-row tile past the batch has no stash tile: nothing of it is stored)
        lane32[st] = 16 * (int)(q & 1) + n + 32 * (g & 1);                // stash lane of (sample, feature-group parity)
    }
    // (the row loads above are older than every weight load: the split above waited for them and, in order, for the constants' DMA)
    S3_STAMP(5, "s_memtime");
    __syncthreads();
    S3_STAMP(1, "s_memtime");

    // Loop-carried besides v1: the PREVIOUS block's coupling inputs v2, t, p of this wave's second-half unit.  The coupling of
    // block b (model.py:411-418: sigmoid, log, (v2 + t) * sigma, operand split, LDS store) runs under the first half of block
    // b+1's S1, whose k-tiles 0, 1 read only the v1 half of the block input; a barrier in the middle of S1 publishes y2 for
    // k-tiles 2, 3.  Block 0 runs it on the input's second half with t = 0, p = 80: sigma = 1 and log = 0 exactly -- the identity.
    f32x4 v1[ST], y2[ST], v2[ST], tt_[ST], pp[ST];
    float lsum[ST];                                                       // running sum of log2(1 + exp(-p)) over all blocks (scaled once, in the epilogue)
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        v1[st] = f32x4{0.f, 0.f, 0.f, 0.f}; y2[st] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[st] = 0.0f;
        v2[st] = xrow[st][1]; tt_[st] = f32x4{0.f, 0.f, 0.f, 0.f}; pp[st] = f32x4{80.f, 80.f, 80.f, 80.f};
    }
    // the deferred coupling of sample tile st; X2: where its operand tiles go (second half of the current block's input)
    auto couple = [&](int st, float* X2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sig, l2;
            lsnf_sigmoid_log2(pp[st][r], sig, l2);
            y2[st][r] = (v2[st][r] + tt_[st][r]) * sig;
            pp[st][r] = sig;                                              // (kept for the stash)
            lsum[st] += l2;
        }
        store_half(X2 + st * L::XT + (HT + nt1) * S3_BTILE_FLOATS, ft1, y2[st], lane);
    };
    // what a call with EXTRAS keeps of block pb once its coupling is done: sigma for the stash, the block's output rows
    auto keep_block = [&](int pb) {
        if (EXTRAS && has1) {
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                if (a.act_saved && tile_ok[st]) {
                    float* act = a.act_saved + (size_t)pb * al.per_block + wtile[st] * al.per_tile;
                    reinterpret_cast<f32x4*>(act + (size_t)nt1 * 1024)[(2 * ft1 + (g >> 1)) * 64 + lane32[st]] = pp[st];
                }
                if (a.z_saved != nullptr && pb + 1 < a.n_blocks && live[st]) {
                    float* zr = a.z_saved + ((size_t)pb * a.B + sample[st]) * a.nz;
                    if (a.vec4 == 4) {
                        store_row_half<HT>(nt1, ft1, v1[st], zr, a.half, g, 4);
                        store_row_half<HT>(HT + nt1, ft1, y2[st], zr, a.half, g, 4);
                    } else {
                        store_row_half<HT>(nt1, ft1, v1[st], zr, a.half, g, a.vec4);
                        store_row_half<HT>(HT + nt1, ft1, y2[st], zr, a.half, g, a.vec4);
                    }
                }
            }
        }
    };
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const float* gblk = a.panels3b + (size_t)blk * C::BLOCK3;
        const bool more = blk + 1 < a.n_blocks;
        const float* gnext = more ? gblk + C::BLOCK3 : gblk;             // last block re-fetches its own panels: no loads under a branch
        // Block input tiles: first half = v1 of the previous block (written by its second S1 half, read by its S2 and by this
        // block's first S1 half), second half = y2 of the previous block (written under this block's first S1 half, read by its
        // second).  One buffer: a writer is always at least one barrier behind the tile's last reader.
        float* Xc = XB;
        float* Xn = XB;
        float* act[ST]; float* hd[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            act[st] = (EXTRAS && a.act_saved && tile_ok[st]) ? a.act_saved + (size_t)blk * al.per_block + wtile[st] * al.per_tile : nullptr;
            hd[st] = (EXTRAS && a.hdump && live[st]) ? a.hdump + (size_t)blk * dl.per_block + sample[st] * (long)a.width : nullptr;   // this sample's row of h1 / h2
        }

        // ---- S1, k-tiles 0..HT-1: v = Wa^T x + ca (model.py:244,268,187); this wave: v1[hu1], v2[hu1].  Under it: the previous
        //      block's coupling; the last k-tile of S4's units (THIS block's fragments) comes in as the carry ----
        if (blk == 1) S3_STAMP(10, "s_memtime");
        f32x4 nv1[ST], nv2[ST];
        {
            const f32x4 b1 = unit_bias(cb + 32 * nt1, ft1, g), b2 = unit_bias(cb + 32 * (HT + nt1), ft1, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) { nv1[st] = b1; nv2[st] = b2; }
        }
        const bf16x8* r1a = unit_ptr<NZT>(gnext, nt1, ft1, lane);
        const bf16x8* r1b = unit_ptr<NZT>(gnext, HT + nt1, ft1, lane);
        {
            const bf16x8* c4t = unit_ptr<WT>(gblk + C::OFF3_S4, nt1, ft1, lane);
            const bf16x8* c4p = unit_ptr<WT>(gblk + C::OFF3_S4, HT + nt1, ft1, lane);
            units_mma_st<NZT, 0, HT, ST, 2, 0, 6, 64>(nv1, nv2, w1a, w1b, r1a, r1b, Xc, L::XT, lane, NoEpi{},
                [&](int q) { if (q < 3) refill_last<WT>(w4t, c4t, q); else refill_last<WT>(w4p, c4p, q - 3); },
                [&](int st) { couple(st, Xc); });
        }
        if (blk > 0) keep_block(blk - 1);            // (EXTRAS only: v1, y2, pp still describe the previous block)
        if (blk == 1) S3_STAMP(11, "s_memtime");
        __syncthreads();                                                  // y2 of the previous block = k-tiles HT.. of this block's input
        if (blk == 1) S3_STAMP(12, "s_memtime");
        // ---- S1, k-tiles HT..NZT-1; v1 -> operand tiles of S2 (= first half of the next block's input) under the last steps ----
        units_mma_st<NZT, HT, NZT, ST, 2, 24, 0, 0>(nv1, nv2, w1a, w1b, r1a, r1b, Xc, L::XT, lane,
            [&](int st) { store_half(Xn + st * L::XT + nt1 * S3_BTILE_FLOATS, ft1, nv1[st], lane); }, [](int) {}, [](int) {});
#pragma unroll
        for (int st = 0; st < ST; ++st) { v1[st] = nv1[st]; v2[st] = nv2[st]; }
        if (wave == 0) {
#pragma unroll
            for (int st = 0; st < ST; ++st) { ell[st] = ell[st] + cb[32 * C::NP + 0]; ell[st] = ell[st] + cb[32 * C::NP + 1]; }   // model.py:273-276, 182,189
        }
        __syncthreads();
        if (blk == 1) S3_STAMP(13, "s_memtime");
        // ---- S2: h1 = relu(actnorm(v1 @ W1)) (model.py:326-328,307) ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 h[ST];
            const f32x4 bh = unit_bias(cb + 32 * (C::P1 + nt), ft, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) h[st] = bh;
            auto epi2 = [&](int st) {
                h[st] = relu4(h[st]);
                store_half(H1B + st * L::HL + nt * S3_BTILE_FLOATS, ft, h[st], lane);
            };
            if (i == 0) {        // carry: the last k-tile of S1's two units (next block's fragments)
                units_mma_st<HT, 0, HT, ST, 1, 28, 6, 0>(h, h, w2[i], w2[i], unit_ptr<HT>(gnext + C::OFF3_S2, nt, ft, lane), nullptr, Xn, L::XT, lane, epi2,
                    [&](int q) { if (q < 3) refill_last<NZT>(w1a, r1a, q); else refill_last<NZT>(w1b, r1b, q - 3); }, [](int) {});
            } else {             // carry: the previous hidden unit's last k-tile
                const bf16x8* cp = unit_ptr<HT>(gnext + C::OFF3_S2, hw[i > 0 ? i - 1 : 0] >> 1, hw[i > 0 ? i - 1 : 0] & 1, lane);
                units_mma_st<HT, 0, HT, ST, 1, 28, 3, 0>(h, h, w2[i], w2[i], unit_ptr<HT>(gnext + C::OFF3_S2, nt, ft, lane), nullptr, Xn, L::XT, lane, epi2,
                    [&](int q) { refill_last<HT>(w2[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
            if (EXTRAS && hasw[i]) {
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    if (hd[st]) store_plain_half(h[st], hd[st] + dl.off_h1, a.width, nt, ft, g, w4);
                    if (act[st]) stash_relu_mask(act[st], al.mask_off, nt, lane32[st], ft, h[st], lane);
                }
            }
        }
        if (blk == 1) S3_STAMP(14, "s_memtime");
        __syncthreads();
        if (blk == 1) S3_STAMP(15, "s_memtime");
        // ---- S3: h2 = relu(actnorm(h1 @ W2)) (model.py:326-328,308) ----
#pragma unroll
        for (int i = 0; i < NU2; ++i) {
            const int nt = hw[i] >> 1, ft = hw[i] & 1;
            f32x4 h[ST];
            const f32x4 bh = unit_bias(cb + 32 * (C::P1 + C::P2 + nt), ft, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) h[st] = bh;
            auto epi3 = [&](int st) {
                h[st] = relu4(h[st]);
                store_half(H2B + st * L::HL + nt * S3_BTILE_FLOATS, ft, h[st], lane);
            };
            // carry: the last k-tile of the hidden unit computed just before (S2's last one, or S3's previous one)
            constexpr int LASTU = NU2 - 1;
            const int cu = i == 0 ? hw[LASTU] : hw[i > 0 ? i - 1 : 0];
            if (i == 0) {
                const bf16x8* cp = unit_ptr<HT>(gnext + C::OFF3_S2, cu >> 1, cu & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 28, 3, 0>(h, h, w3[i], w3[i], unit_ptr<WT>(gnext + C::OFF3_S3, nt, ft, lane), nullptr, H1B, L::HL, lane, epi3,
                    [&](int q) { refill_last<HT>(w2[LASTU], cp, q); }, [](int) {});
            } else {
                const bf16x8* cp = unit_ptr<WT>(gnext + C::OFF3_S3, cu >> 1, cu & 1, lane);
                units_mma_st<WT, 0, WT, ST, 1, 28, 3, 0>(h, h, w3[i], w3[i], unit_ptr<WT>(gnext + C::OFF3_S3, nt, ft, lane), nullptr, H1B, L::HL, lane, epi3,
                    [&](int q) { refill_last<WT>(w3[i > 0 ? i - 1 : 0], cp, q); }, [](int) {});
            }
            if (EXTRAS && hasw[i]) {
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    if (hd[st]) store_plain_half(h[st], hd[st] + dl.off_h2, a.width, nt, ft, g, w4);
                    if (act[st]) stash_relu_mask(act[st], al.mask_off, WT + nt, lane32[st], ft, h[st], lane);
                }
            }
        }
        if (blk == 1) S3_STAMP(16, "s_memtime");
        __syncthreads();
        if (blk == 1) S3_STAMP(17, "s_memtime");
        // ---- S4: shift t / pre-sigmoid p (model.py:347-349,411-413); the coupling itself (:414-418) rides under the next S1 ----
        {
            const f32x4 bt = unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + nt1), ft1, g);
            const f32x4 bp = unit_bias(cb + 32 * (C::P1 + C::P2 + C::P3 + HT + nt1), ft1, g);
#pragma unroll
            for (int st = 0; st < ST; ++st) { tt_[st] = bt; pp[st] = bp; }
        }
        {   // carry: the last k-tile of S3's last hidden unit (next block's fragments)
            constexpr int LASTU = NU2 - 1;
            const bf16x8* c3 = unit_ptr<WT>(gnext + C::OFF3_S3, hw[LASTU] >> 1, hw[LASTU] & 1, lane);
            units_mma_st<WT, 0, WT, ST, 2, 0, 3, 0>(tt_, pp, w4t, w4p, unit_ptr<WT>(gnext + C::OFF3_S4, nt1, ft1, lane), unit_ptr<WT>(gnext + C::OFF3_S4, HT + nt1, ft1, lane), H2B, L::HL, lane,
                NoEpi{}, [&](int q) { refill_last<WT>(w3[LASTU], c3, q); }, [](int) {});
        }
        if (blk == 1) S3_STAMP(18, "s_memtime");
        // (no barrier here: the next block's first S1 half reads tiles that were published before this block's S2)
        if (blk == 1) S3_STAMP(20, "s_memtime");
    }
    // the last block's coupling (no S1 follows); its operand tiles are not needed (nobody reads X any more)
#pragma unroll
    for (int st = 0; st < ST; ++st) couple(st, XB);
    keep_block(a.n_blocks - 1);
    S3_STAMP(40, "s_memtime");

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        float ss = 0.0f, part = 0.0f;
        if (has1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += v1[st][r] * v1[st][r] + y2[st][r] * y2[st][r];
            part = lsum[st];                                              // this wave's 16 features of sum_j log2(1 + exp(-p_j)), all blocks
            if (live[st]) {
                float* zr = a.z_out + sample[st] * (long)a.nz;
                if (a.vec4 == 4) {
                    store_row_half<HT>(nt1, ft1, v1[st], zr, a.half, g, 4);
                    store_row_half<HT>(HT + nt1, ft1, y2[st], zr, a.half, g, 4);
                } else {
                    store_row_half<HT>(nt1, ft1, v1[st], zr, a.half, g, a.vec4);
                    store_row_half<HT>(HT + nt1, ft1, y2[st], zr, a.half, g, a.vec4);
                }
            }
        }
        ss = group_sum(ss);
        const float el = ell[st] + -0.6931471805599453f * group_sum(part);   // model.py:418
        if (g == 0) { RED[((st * 4 + wave) * 16 + n) * 2] = ss; RED[((st * 4 + wave) * 16 + n) * 2 + 1] = el; }
    }
    __syncthreads();
    if (wave == 0) {
        double dl = 0.0, dd = 0.0;
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            float s2 = 0.0f, el = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s2 += RED[((st * 4 + w) * 16 + n) * 2]; el += RED[((st * 4 + w) * 16 + n) * 2 + 1]; }
            const float ll = (-0.5f * s2 + 1.8378770664093453f) + el;
            if (live[st] && g == 0) {
                a.logdet_out[sample[st]] = el;
                if (a.ll_out) a.ll_out[sample[st]] = ll;
                dl += (double)ll; dd += (double)el;
            }
        }
        if (a.stats) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
            lsnf_publish_stats(a.stats, dl, dd, a.B, lane);       // (wave-level protocol: all 64 lanes of wave 0)
        }
    }
    S3_STAMP(41, "s_memtime");
    S3_STAMP(51, "s_memrealtime");
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

template <class C, int ST>
hipError_t launch_small3_fwd_st(const Small3Args& a, hipStream_t stream) {
    if constexpr ((size_t)(Small3Lds<C, ST>::L_CONST + C::CONST_FLOATS) * sizeof(float) > 160 * 1024 || (ST == 4 && C::WT > 2)) {
        return hipErrorInvalidValue;                 // (this shape cannot fit for any depth: not instantiated)
    } else {
        const size_t lds = ((size_t)Small3Lds<C, ST>::L_CONST + (size_t)a.n_blocks * C::CONST_FLOATS) * sizeof(float);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        const bool extras = a.z_saved != nullptr || a.act_saved != nullptr || a.hdump != nullptr;
        auto kern = extras ? lsnf_small3_fwd_kernel<C, ST, true> : lsnf_small3_fwd_kernel<C, ST, false>;
        static unsigned long long lds_ok[2] = {0, 0};
        if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok[extras]); e != hipSuccess) return e;
        const unsigned grid = (unsigned)((a.B + ST * S3_SAMPLES - 1) / (ST * S3_SAMPLES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
        return hipGetLastError();
    }
}
// Rows per workgroup by batch size: 16 while one round of workgroups covers the batch (<= 256 CUs x 16 rows), then 32, then 64
// -- the weight stream per workgroup is the same, so a second round (or a second workgroup per CU) costs a whole stream
// while a second sample tile costs its MFMAs only.  LSNF_SMALL3_ST (1 / 2 / 4) forces a shape (experiments, tests).
template <class C>
hipError_t launch_small3_fwd(const Small3Args& a, hipStream_t stream) {
    static const char* env = getenv("LSNF_SMALL3_ST");
    const bool extras = a.z_saved != nullptr || a.act_saved != nullptr || a.hdump != nullptr;
    // (the plain 32-row form runs two workgroups per CU: 512 of them cover 16 384 rows in one round)
    int st = env ? atoi(env) : (a.B <= 256 * 16 ? 1 : ((a.B <= 256 * 32 || (LSNF_SMALL3_WAVES2 && !extras && C::WT <= 2 && a.B <= 512 * 32)) ? 2 : 4));
    hipError_t e = hipErrorInvalidValue;
    if (st >= 4) e = launch_small3_fwd_st<C, 4>(a, stream);
    if (e == hipErrorInvalidValue && st >= 2) e = launch_small3_fwd_st<C, 2>(a, stream);     // (the larger shape did not fit into LDS)
    if (e == hipErrorInvalidValue) e = launch_small3_fwd_st<C, 1>(a, stream);
    return e;
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
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    { extern unsigned long long* g_lsnf_stamps;
      if (!g_lsnf_stamps) { if (hipMalloc(&g_lsnf_stamps, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) g_lsnf_stamps = nullptr; }
      a.stamps = g_lsnf_stamps; }
#endif
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
