// lsnf_device.h -- device-side building blocks shared by the forward / reverse / backward kernels.
// gfx950 (CDNA4) only: wave64, v_mfma_f32_32x32x2_f32, LDS-DMA (global_load_lds_dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "lsnf_layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LSNF_WG_THREADS 256          // 4 waves, one per SIMD; 2 workgroups co-resident per CU
#define LSNF_WG_WAVES 4
#define LSNF_WG_SAMPLES (LSNF_WG_WAVES * 32)

#define LSNF_AS1 __attribute__((address_space(1)))
#define LSNF_AS3 __attribute__((address_space(3)))

// Host side: kernels that use more than 64 KiB of dynamic LDS need the attribute set once PER DEVICE of the process.
static inline hipError_t lsnf_allow_big_lds(const void* kernel, unsigned long long* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((*done_mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) *done_mask |= 1ull << dev;      // benign race: the call is idempotent
    return hipSuccess;
}

// Compile-time geometry of one coupling block for a kernel instantiation (HT = tiles of nz/2, WT = tiles of f_width):
// panel counts / k-tiles of the forward stages S1..S4 and of the backward stages B4..B1, and the float offsets of
// their packed panels inside a block's forward / backward stream (layout produced by lsnf_prep.hip).
template <int HT_, int WT_>
struct LsnfStackCfg {
    static constexpr int HT = HT_, WT = WT_, NZT = 2 * HT_;
    static constexpr int P1 = NZT, P2 = WT, P3 = WT, P4 = 2 * HT, NP = P1 + P2 + P3 + P4;   // n-tiles per forward stage
    static constexpr int KT1 = NZT, KT2 = HT, KT3 = WT, KT4 = WT;
    static constexpr int OFF_S2 = LSNF_FRAG_FLOATS * P1 * KT1;
    static constexpr int OFF_S3 = OFF_S2 + LSNF_FRAG_FLOATS * P2 * KT2;
    static constexpr int OFF_S4 = OFF_S3 + LSNF_FRAG_FLOATS * P3 * KT3;
    static constexpr int FWD_BLOCK = OFF_S4 + LSNF_FRAG_FLOATS * P4 * KT4;
    static constexpr int FWD_CONST = 32 * NP + 32;             // biases of S1..S4, then [sum 3 logs, log|det W|, pad]
    static constexpr int INV_BLOCK = LSNF_FRAG_FLOATS * NZT * NZT;
    static constexpr int INV_CONST = 32 * NZT;
    static constexpr int OFF_B4 = 0;
    static constexpr int OFF_B3 = OFF_B4 + LSNF_FRAG_FLOATS * WT * 2 * HT;
    static constexpr int OFF_B2 = OFF_B3 + LSNF_FRAG_FLOATS * WT * WT;
    static constexpr int OFF_B1 = OFF_B2 + LSNF_FRAG_FLOATS * HT * WT;
    static constexpr int BWD_BLOCK = OFF_B1 + LSNF_FRAG_FLOATS * NZT * NZT;
    static constexpr int MAXKT = (NZT > WT ? NZT : WT);
    static constexpr int SLOT = 2 * MAXKT * LSNF_FRAG_FLOATS;  // LDS floats of one panel PAIR (throughput kernels)
};

// In-kernel batch sums (stats argument of lsnf_forward): called by ALL 64 lanes of ONE wave per workgroup, lane 0 holding
// the workgroup's partial sums.  No fences (a release fence here would flush this workgroup's freshly written z_out lines:
// +2..6 us per workgroup); ordering comes from RETURNING atomics, whose values come back only after the operation has been
// performed at the memory side, and from making every later step depend on them:
//   1. lane 0 adds the partial sums into sub-accumulator s = workgroup % 64 (stats[8 + 4 s ..]: sum ll, sum logdet, ticket);
//   2. it draws the sub-accumulator's ticket (grids of more than 64 workgroups; otherwise every workgroup has its own slot);
//   3. the last arrival of a sub-accumulator draws the launch-wide ticket (stats[2]);
//   4. the last of those reads all 64 sub-accumulators -- one atomic exchange with zero per lane, which re-arms them for the
//      next launch as well (a plain load could be served by this XCD's L2, which is not coherent with the other XCDs) --
//      and publishes stats[4..6].
// Atomics on ONE address serialise at the memory side (~40 ns each: 40 us for the 1 024 workgroups of a 16 384-row latency
// launch with a single accumulator, tools/shard_times.py); the longest chain here is four dependent round trips.
#define LSNF_STATS_SLOTS 64
__device__ __forceinline__ void lsnf_publish_stats(double* stats, double sum_ll, double sum_logdet, int rows, int lane) {
    const unsigned grid = gridDim.x, wg = blockIdx.x;
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(&stats[2]);
    int last = 0;
    if (lane == 0) {
        const unsigned s = wg % LSNF_STATS_SLOTS;
        double* sub = stats + 8 + 4 * s;
        const double r0 = atomicAdd(&sub[0], sum_ll);
        const double r1 = atomicAdd(&sub[1], sum_logdet);
        unsigned long long inc = 1ull;
        asm volatile("" : "+v"(inc) : "v"(r0), "v"(r1));          // the ticket depends on the adds having been performed
        bool slot_done = true;
        if (grid > LSNF_STATS_SLOTS) {
            const unsigned long long n_s = (grid - s + LSNF_STATS_SLOTS - 1) / LSNF_STATS_SLOTS;       // workgroups of this slot
            unsigned long long* st = reinterpret_cast<unsigned long long*>(&sub[2]);
            slot_done = atomicAdd(st, inc) == n_s - 1;
            if (slot_done) atomicExch(st, 0ull);                   // re-arm, fire and forget
            inc = 1ull;
        }
        if (slot_done) {
            const unsigned long long slots = grid > LSNF_STATS_SLOTS ? LSNF_STATS_SLOTS : grid;
            if (atomicAdd(ticket, inc) == slots - 1) { last = 1; atomicExch(ticket, 0ull); }
        }
    }
    if (__builtin_amdgcn_readfirstlane(last)) {                    // wave-uniform (lane 0 is the first lane)
        unsigned long long* sub = reinterpret_cast<unsigned long long*>(stats + 8 + 4 * lane);
        double fl = __longlong_as_double((long long)atomicExch(&sub[0], 0ull));
        double fd = __longlong_as_double((long long)atomicExch(&sub[1], 0ull));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { fl += __shfl_xor(fl, o, 64); fd += __shfl_xor(fd, o, 64); }
        if (lane == 0) { stats[4] = fl; stats[5] = fd; stats[6] = (double)rows; }
    }
}

// feature offset inside a 32-tile of accumulator register r on lane-half h
__device__ __forceinline__ constexpr int lsnf_feat(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- LDS-DMA of one weight panel (KT KiB*4) by the 4 waves of the workgroup -------------------
// Each wave-instruction moves 1 KiB (64 lanes x 16 B), destination = wave-uniform base + lane*16.
template <int KT, int NW = LSNF_WG_WAVES>
__device__ __forceinline__ void lsnf_issue_panel(const float* __restrict__ gsrc, float* lbuf, int wave, int lane) {
#ifdef LSNF_ABLATE_DMA   // timing diagnostic only (wrong numbers): prices the L2 -> LDS weight stream and its power
    return;
#endif
    constexpr int PIECES = 4 * KT;                 // 1 KiB pieces of this panel (pair)
    constexpr int PER_WAVE = (PIECES + NW - 1) / NW;
#pragma unroll
    for (int s = 0; s < PER_WAVE; ++s) {
        const int seg = s * NW + wave;             // 1 KiB segment index
        if (PIECES % NW == 0 || seg < PIECES) {    // wave-uniform
            const float* g = gsrc + seg * 256 + lane * 4;
            float* l = lbuf + seg * 256;
            __builtin_amdgcn_global_load_lds((const LSNF_AS1 void*)g, (LSNF_AS3 void*)l, 16, 0, 0);
        }
    }
}

// Wait for this wave's outstanding LDS-DMA (and any other VMEM), then workgroup barrier:
// after it every wave's part of the awaited panel is visible and every wave has finished
// reading the panel that was consumed before.
__device__ __forceinline__ void lsnf_panel_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef LSNF_ABLATE_BARRIER   // timing diagnostic only (racy): prices the per-panel workgroup barrier
    __syncthreads();
#endif
}

// ---- one panel of MFMAs: acc(32 features x 32 samples) += W_panel^T * in ----------------------
// lbuf: panel in LDS (fragment order), in[kt]: activations tile kt (B operand), acc: C/D.
// Software-pipelined per k-tile: the four ds_read_b128 of k-tile kt+1 are issued BEFORE the 16 MFMAs of
// k-tile kt (two sets of fragment registers), so that hipcc's lgkmcnt(0) before the next tile's first
// MFMA finds the reads long complete.  The sched_group_barriers pin that order (left alone hipcc emits
// read -> lgkmcnt(0) -> 4 MFMA, fully serialised: measured 27 % slower).
template <int KT>
__device__ __forceinline__ void lsnf_panel_mma(f32x16& acc, const f32x16* in, const float* lbuf, int lane) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(lbuf) + lane;
    __builtin_amdgcn_sched_barrier(0);                       // nothing from before drifts into the pipeline
    f32x4 w0 = wp[0], w1 = wp[64], w2 = wp[128], w3 = wp[192];
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // k-tile 0: its four fragment reads
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        f32x4 n0 = w0, n1 = w1, n2 = w2, n3 = w3;
        if (kt + 1 < KT) {                                   // whole next k-tile in flight under 16 MFMAs
            n0 = wp[((kt + 1) * 4 + 0) * 64]; n1 = wp[((kt + 1) * 4 + 1) * 64];
            n2 = wp[((kt + 1) * 4 + 2) * 64]; n3 = wp[((kt + 1) * 4 + 3) * 64];
        }
#define LSNF_MFMA4(W, G)                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[0], in[kt][4 * G + 0], acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[1], in[kt][4 * G + 1], acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[2], in[kt][4 * G + 2], acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[3], in[kt][4 * G + 3], acc, 0, 0, 0);
        LSNF_MFMA4(w0, 0) LSNF_MFMA4(w1, 1) LSNF_MFMA4(w2, 2) LSNF_MFMA4(w3, 3)
#undef LSNF_MFMA4
        if (kt + 1 < KT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads (prefetch) first
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                   // then this k-tile's 16 MFMA
        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
}

// accumulator initialised with the per-feature bias (bias block: [h][r], 32 floats per n-tile)
__device__ __forceinline__ f32x16 lsnf_bias_init(const float* cst, int h) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + h * 16);
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = b[q];
        a[4 * q + 0] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
    }
    return a;
}

__device__ __forceinline__ f32x16 lsnf_zero16() {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
    return a;
}

__device__ __forceinline__ f32x16 lsnf_relu16(f32x16 a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
    return a;
}

// sig = sigmoid(p) and l2 = log2(1 + exp(-p)) = -log2(sigmoid(p)) on the hardware transcendentals
// (v_exp_f32 / v_rcp_f32 / v_log_f32, <= 1 ulp each): 7 VALU instructions.  p is clamped to [-80, 80] (v_med3) so
// that exp never overflows; |pre-sigmoid| beyond 80 does not occur (sigmoid saturates in fp32 at ~17), and the
// clamp changes log(sigmoid) only where it is already < -80.  Absolute error of l2 <= ~1.5e-7 (rounding of 1+e),
// i.e. ~1e-8 relative on a log-prob of magnitude 80 after summing a sample's 320 terms.
__device__ __forceinline__ void lsnf_sigmoid_log2(float p, float& sig, float& l2) {
    const float pc = __builtin_amdgcn_fmed3f(p, -80.0f, 80.0f);
    const float e = __builtin_amdgcn_exp2f(pc * -1.4426950408889634f);    // exp(-p)
    const float d = 1.0f + e;
    sig = __builtin_amdgcn_rcpf(d);
    l2 = __builtin_amdgcn_logf(d);                                         // v_log_f32 is log2
}

// sigma = sigmoid(p), lsig = log(sigmoid(p)), both stable for any finite p.
//   reference: scale = sigmoid(h[:,1::2] + 2) (model.py:413; the +2 is folded into the bias),
//              log(scale) (model.py:418)
__device__ __forceinline__ void lsnf_sigmoid_logsig(float p, float& sig, float& lsig) {
#ifdef LSNF_ABLATE_EPILOGUE   // timing diagnostic only (wrong numbers): prices the transcendental epilogue
    sig = p * 0.25f + 0.5f; lsig = p; return;
#endif
#ifdef LSNF_OCML_MATH               // reference build of the epilogue on OCML expf/log1pf (slow, ~110 VALU/element)
    const float a = fabsf(p);
    const float e = expf(-a);
    const float r = 1.0f / (1.0f + e);
    const float l = log1pf(e);
    sig = (p >= 0.0f) ? r : e * r;
    lsig = fminf(p, 0.0f) - l;
#else
    float l2;
    lsnf_sigmoid_log2(p, sig, l2);
    lsig = -0.6931471805599453f * l2;
#endif
}

// Lane exchanges on the vector ALU (gfx950: v_permlane16_swap / v_permlane32_swap; __shfl_xor goes through the LDS crossbar:
// a ~100-cycle round trip).  swap(v, v) returns the pair (r0, r1) in which every lane holds its own value in one and its
// partner's (lane ^ 16 resp. lane ^ 32) in the other: own (+ or |) partner = r0 (+ or |) r1 on every lane.
__device__ __forceinline__ float lsnf_pair_add16(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float lsnf_pair_add32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ unsigned lsnf_pair_or32(unsigned u) {
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return (unsigned)r[0] | (unsigned)r[1];
}
// sum of a value held by lanes l and l+32 (the two feature half-groups of one sample)
__device__ __forceinline__ float lsnf_pair_sum(float v) { return lsnf_pair_add32(v); }

// ---- latent rows <-> split-pad register tiles -------------------------------------------------
// row: sample index (already clamped to [0,B)), tile t of the split-pad row: x[r] <- feature nat(32*t + o(r,h)).
// vw = vector width of the global accesses: 4 (half % 4 == 0, 16-byte aligned rows), 2 (half even, 8-byte aligned:
// e.g. nz = 100) or 1.  A 4-feature group never straddles the valid / padded boundary at its own granularity.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int HT>
__device__ __forceinline__ f32x16 lsnf_load_tile(int t, const float* __restrict__ zr, int half, int h, int vw) {
    f32x16 x;
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * tt + 8 * g + 4 * h;
        const int col0 = hh * half + f0;
        if (vw == 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f0 < half) v = *reinterpret_cast<const f32x4*>(zr + col0);
            x[4 * g + 0] = v[0]; x[4 * g + 1] = v[1]; x[4 * g + 2] = v[2]; x[4 * g + 3] = v[3];
        } else if (vw == 2) {
            f32x2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
            if (f0 < half) v0 = *reinterpret_cast<const f32x2*>(zr + col0);
            if (f0 + 2 < half) v1 = *reinterpret_cast<const f32x2*>(zr + col0 + 2);
            x[4 * g + 0] = v0[0]; x[4 * g + 1] = v0[1]; x[4 * g + 2] = v1[0]; x[4 * g + 3] = v1[1];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[4 * g + j] = (f0 + j < half) ? zr[col0 + j] : 0.0f;
        }
    }
    return x;
}

template <int HT>
__device__ __forceinline__ void lsnf_store_tile(int t, const f32x16& x, float* __restrict__ zr, int half, int h, int vw) {
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * tt + 8 * g + 4 * h;
        const int col0 = hh * half + f0;
        if (vw == 4) {
            if (f0 < half) {
                f32x4 v = {x[4 * g + 0], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                *reinterpret_cast<f32x4*>(zr + col0) = v;
            }
        } else if (vw == 2) {
            if (f0 < half) { f32x2 v = {x[4 * g + 0], x[4 * g + 1]}; *reinterpret_cast<f32x2*>(zr + col0) = v; }
            if (f0 + 2 < half) { f32x2 v = {x[4 * g + 2], x[4 * g + 3]}; *reinterpret_cast<f32x2*>(zr + col0 + 2) = v; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < half) zr[col0 + j] = x[4 * g + j];
        }
    }
}

template <int HT>
__device__ __forceinline__ void lsnf_load_rows(f32x16* x, const float* __restrict__ z, long row, int nz, int half,
                                               int h, int vec4) {
    const float* zr = z + row * (long)nz;
#pragma unroll
    for (int t = 0; t < 2 * HT; ++t) x[t] = lsnf_load_tile<HT>(t, zr, half, h, vec4);
}

template <int HT>
__device__ __forceinline__ void lsnf_store_rows(const f32x16* x, float* __restrict__ z, long row, int nz, int half,
                                                int h, int vec4) {
    float* zr = z + row * (long)nz;
#pragma unroll
    for (int t = 0; t < 2 * HT; ++t) lsnf_store_tile<HT>(t, x[t], zr, half, h, vec4);
}

// ---- double-buffered weight-panel pipeline with a run-time buffer parity -------------------------
// acquire<KT_NEXT>(next): (1) wait for my LDS-DMA + workgroup barrier: the panel issued one acquire
// ago is now complete in buf[cur] and every wave is done with buf[cur^1]; (2) start the LDS-DMA of
// the NEXT panel (KT_NEXT k-tiles at `next`, or nothing if next == nullptr) into buf[cur^1];
// (3) return buf[cur] for the MFMAs and flip.  The very first panel is issued with prime<KT>().
template <int NW>
struct LsnfPipeT {
    float* buf0;
    int slot;     // floats per buffer
    int cur;      // wave-uniform
    int wave, lane;

    template <int KT>
    __device__ __forceinline__ void prime(const float* src) {
        lsnf_issue_panel<KT, NW>(src, buf0, wave, lane);
        cur = 0;
    }
    template <int KT_NEXT>
    __device__ __forceinline__ const float* acquire(const float* next) {
        lsnf_panel_barrier();
        if (next != nullptr) lsnf_issue_panel<KT_NEXT, NW>(next, buf0 + (cur ^ 1) * slot, wave, lane);
        const float* ready = buf0 + cur * slot;
        cur ^= 1;
        return ready;
    }
};
using LsnfPipe = LsnfPipeT<LSNF_WG_WAVES>;

template <int N, class F, int I = 0>
__device__ __forceinline__ void lsnf_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lsnf_static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// ---- two n-tiles at once: two independent accumulator chains sharing the B operand ---------------------
// lbuf holds tile 0's KT k-tiles followed by tile 1's (two consecutive panels of the packed stream).
// Per half k-tile: 4 ds_read_b128 (2 fragment groups x 2 tiles) are prefetched under the previous 16 MFMAs.
template <int KT>
__device__ __forceinline__ void lsnf_panel_mma2(f32x16& acc0, f32x16& acc1, const f32x16* in, const float* lbuf, int lane) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(lbuf) + lane;
    constexpr int T1 = KT * 4 * 64;                          // f32x4 offset of tile 1's panel
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a0 = wp[0], a1 = wp[64], b0 = wp[T1], b1 = wp[T1 + 64];
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int H = 0; H < 2 * KT; ++H) {                       // half k-tiles: groups 2H, 2H+1
        const int kt = H >> 1, g0 = (H & 1) * 2;
        f32x4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
#ifndef LSNF_ABLATE_LDSREAD   // timing diagnostic only (wrong numbers): prices the LDS fragment reads / their power
        if (H + 1 < 2 * KT) {
            na0 = wp[(2 * H + 2) * 64]; na1 = wp[(2 * H + 3) * 64];
            nb0 = wp[T1 + (2 * H + 2) * 64]; nb1 = wp[T1 + (2 * H + 3) * 64];
        }
#endif
#define LSNF_MFMA2(WA, WB, G)                                                                         \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[0], in[kt][4 * (G) + 0], acc0, 0, 0, 0);       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[0], in[kt][4 * (G) + 0], acc1, 0, 0, 0);       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[1], in[kt][4 * (G) + 1], acc0, 0, 0, 0);       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[1], in[kt][4 * (G) + 1], acc1, 0, 0, 0);       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[2], in[kt][4 * (G) + 2], acc0, 0, 0, 0);       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[2], in[kt][4 * (G) + 2], acc1, 0, 0, 0);       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[3], in[kt][4 * (G) + 3], acc0, 0, 0, 0);       \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[3], in[kt][4 * (G) + 3], acc1, 0, 0, 0);
        LSNF_MFMA2(a0, b0, g0) LSNF_MFMA2(a1, b1, g0 + 1)
#undef LSNF_MFMA2
        if (H + 1 < 2 * KT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

// ---- one GEMM stage: out[t] = post(init(t) + W_t^T in), t < NT, streamed as panel PAIRS -----------------
// gsrc: this stage's packed panels (n-tile major, KT k-tiles each).  gnext/NEXT_KTC: first panel (pair) of
// whatever follows this stage (NEXT_KTC = its k-tiles x its tile count; gnext == nullptr: nothing follows).
template <int NT, int KT, int NEXT_KTC, class Pipe, class Init, class Post>
__device__ __forceinline__ void lsnf_gemm_stage(Pipe& pipe, const float* gsrc, const float* gnext, f32x16* out,
                                                const f32x16* in, Init&& init, Post&& post) {
    constexpr int NSP = (NT + 1) / 2;
    lsnf_static_for<NSP>([&](auto qc) {
        constexpr int q = decltype(qc)::value, t0 = 2 * q, cnt = (NT - t0 >= 2) ? 2 : 1;
        const float* lb;
        if constexpr (q + 1 < NSP) {
            constexpr int cn = (NT - (t0 + 2) >= 2) ? 2 : 1;
            lb = pipe.template acquire<KT * cn>(gsrc + (t0 + 2) * KT * LSNF_FRAG_FLOATS);
        } else {
            lb = pipe.template acquire<NEXT_KTC>(gnext);
        }
        out[t0] = init(t0);
        if constexpr (cnt == 2) {
            out[t0 + 1] = init(t0 + 1);
            lsnf_panel_mma2<KT>(out[t0], out[t0 + 1], in, lb, pipe.lane);
            out[t0 + 1] = post(out[t0 + 1], t0 + 1);
        } else {
            lsnf_panel_mma<KT>(out[t0], in, lb, pipe.lane);
        }
        out[t0] = post(out[t0], t0);
    });
}
// k-tiles x tiles of the FIRST panel pair of a stage with NT tiles of KT k-tiles
__device__ __forceinline__ constexpr int lsnf_first_ktc(int NT, int KT) { return KT * (NT >= 2 ? 2 : 1); }

// activation stash helpers (lsnf_layout.h, LsnfActLayout): sigma tile / relu-mask word of one 32-sample tile
__device__ __forceinline__ void lsnf_act_store_sigma(float* tile_base, int t, const f32x16& sg, int lane) {
    f32x4* p = reinterpret_cast<f32x4*>(tile_base + (size_t)t * 1024) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) { f32x4 v = {sg[4 * g], sg[4 * g + 1], sg[4 * g + 2], sg[4 * g + 3]}; p[g * 64] = v; }
}
__device__ __forceinline__ f32x16 lsnf_act_load_sigma(const float* tile_base, int t, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(tile_base + (size_t)t * 1024) + lane;
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) { const f32x4 v = p[g * 64]; a[4 * g] = v[0]; a[4 * g + 1] = v[1]; a[4 * g + 2] = v[2]; a[4 * g + 3] = v[3]; }
    return a;
}
__device__ __forceinline__ unsigned* lsnf_act_mask_ptr(float* tile_base, size_t mask_off, int idx, int lane) {
    return reinterpret_cast<unsigned*>(tile_base + mask_off) + idx * 64 + lane;
}
__device__ __forceinline__ const unsigned* lsnf_act_mask_ptr(const float* tile_base, size_t mask_off, int idx, int lane) {
    return reinterpret_cast<const unsigned*>(tile_base + mask_off) + idx * 64 + lane;
}

// ---- counter-based Gaussian noise for the fused Langevin update (train.py:326, `s * randn_like(z)`) ----
// noise(row, col), col = hh*half + f (hh: which half of the latent, f: feature within it), is normal number (f & 3)
// of the four made from Philox4x32-10 with
//   counter = ( (hh << 16) | (f >> 2),  row & 0xffffffff,  offset & 0xffffffff,  (offset >> 32) ^ (row >> 32) )
//   key     = ( seed & 0xffffffff, seed >> 32 ),            row = global row index (row0 + local row)
// Box-Muller on (x0,x1) -> normals 0,1 and (x2,x3) -> normals 2,3, u = ((x >> 8) + 0.5) * 2^-24.
// A pure function of (seed, offset, row, col): independent of the kernel family, of the batch sharding and of B.
__device__ __forceinline__ void lsnf_philox4x32_10(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void lsnf_box_muller(unsigned xa, unsigned xb, float& n0, float& n1) {
    const float u1 = ((float)(xa >> 8) + 0.5f) * 5.9604644775390625e-08f;      // (0, 1)
    const float u2 = ((float)(xb >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), v_log_f32 = log2
    n0 = r * __builtin_amdgcn_cosf(u2);                                         // v_cos/v_sin take revolutions
    n1 = r * __builtin_amdgcn_sinf(u2);
}
struct LsnfRngState { unsigned k0, k1, c2, c3hi; int on; };
template <int HT>
__device__ __forceinline__ f32x16 lsnf_noise_tile(int t, unsigned long long grow, int half, int h, const LsnfRngState& st) {
    f32x16 x;
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * tt + 8 * g + 4 * h;
        unsigned c0 = ((unsigned)hh << 16) | (unsigned)(f0 >> 2), c1 = (unsigned)grow, c2 = st.c2, c3 = st.c3hi ^ (unsigned)(grow >> 32);
        lsnf_philox4x32_10(c0, c1, c2, c3, st.k0, st.k1);
        float n0, n1, n2, n3;
        lsnf_box_muller(c0, c1, n0, n1);
        lsnf_box_muller(c2, c3, n2, n3);
        x[4 * g + 0] = n0; x[4 * g + 1] = n1; x[4 * g + 2] = n2; x[4 * g + 3] = n3;
    }
    return x;
}

// bit r of the result = (a[r] > 0): relu mask of one tile, for the backward pass
__device__ __forceinline__ unsigned lsnf_posmask16(const f32x16& a) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (a[r] > 0.0f ? 1u : 0u) << r;
    return m;
}
__device__ __forceinline__ f32x16 lsnf_apply_mask16(f32x16 a, unsigned m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ((m >> r) & 1u) ? a[r] : 0.0f;
    return a;
}
