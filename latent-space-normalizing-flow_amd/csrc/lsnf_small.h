// lsnf_small.h -- building blocks of the SMALL-BATCH kernels (lsnf_small_fwd.hip, lsnf_small_bwd.hip).
//
// The throughput kernels give each wave 32 samples and ALL features: 2560 dependent-chain MFMAs per wave and
// forward, so a batch of <= 128 rows runs on one CU for ~90 us.  Here a workgroup owns only 32 samples and its
// 4 waves split every GEMM stage between them -- by output tile (NT >= 4) or by output tile x K range
// (NT = 2: K split in two, NT = 1: in four) -- so the serial MFMA chain per stage is 16..64 instead of
// 64..256.  Stage outputs travel wave -> LDS -> wave as accumulator-layout tiles (partial sums when K is split;
// the consumer adds them and applies the activation), which is exactly the B-operand layout the next stage
// needs.  Weights are not shared between waves any more, so they skip LDS: each wave loads the fragments of
// ITS panel slice straight from L2 into VGPRs (lane-linear 16 B per lane), one stage ahead of their use.
#pragma once
#include <type_traits>
#include "lsnf_device.h"

#define LSNF_SMALL_SAMPLES 32
#define LSNF_TILE_FLOATS 1024   // one activation tile in LDS: [g(4)][lane(64)][4 floats]

// how a stage of NT output tiles and KT k-tiles is spread over 4 waves (FORCE_KS > 0 overrides the K split)
template <int NT, int KT, int FORCE_KS = 0>
struct SmallSplit {
    static constexpr int AUTO = (NT >= 4) ? 1 : (NT == 2 ? (KT >= 2 ? 2 : 1) : (NT == 1 ? (KT >= 4 ? 4 : (KT >= 2 ? 2 : 1)) : 1));
    static constexpr int KS = FORCE_KS > 0 ? FORCE_KS : AUTO;
    static constexpr int KTL = KT / KS;                  // k-tiles per unit
    static constexpr int UNITS = NT * KS;                // units of work (<= 4: one per wave)
    static_assert(KT % KS == 0, "K split must divide KT");
    static_assert(UNITS <= 4, "one unit per wave");
};

// tile in accumulator layout <-> LDS tile
__device__ __forceinline__ void small_store_tile(float* lds_tile, const f32x16& a, int lane) {
    f32x4* p = reinterpret_cast<f32x4*>(lds_tile) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) { f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}; p[g * 64] = v; }
}
__device__ __forceinline__ f32x16 small_load_tile(const float* lds_tile, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(lds_tile) + lane;
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) { const f32x4 v = p[g * 64]; a[4 * g] = v[0]; a[4 * g + 1] = v[1]; a[4 * g + 2] = v[2]; a[4 * g + 3] = v[3]; }
    return a;
}
// sum of KS partial tiles (consecutive in LDS), optional relu
template <int KS, bool RELU>
__device__ __forceinline__ f32x16 small_gather_tile(const float* lds_tiles, int lane) {
    f32x16 a = small_load_tile(lds_tiles, lane);
#pragma unroll
    for (int s = 1; s < KS; ++s) {
        const f32x16 b = small_load_tile(lds_tiles + s * LSNF_TILE_FLOATS, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += b[r];
    }
    if (RELU) a = lsnf_relu16(a);
    return a;
}
// a[r] = (gate[r] > 0) ? a[r] : 0      (relu backward, model.py:307-308)
__device__ __forceinline__ f32x16 small_gate16(f32x16 a, const f32x16& gate) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = gate[r] > 0.0f ? a[r] : 0.0f;
    return a;
}

// weight fragments of one unit (KTL k-tiles of one n-tile) held in registers
template <int KTL>
struct SmallFrags { f32x4 w[KTL * 4]; };

// a fetch in progress: pointer to this wave's slice + the registers it lands in.  The loads are issued by the
// stage that runs meanwhile (SmallStage::run(..., fetch)), one or two after each group of 4 MFMAs: issued in one
// burst at the start of a stage they hold the wave's MFMA chain back for the ~16 cycles per KiB the CU's vector
// memory path needs (64 B/clk: 2000+ cycles per block, ~15 % of the kernel at batch 100).
template <int KTL>
struct SmallFetch {
    static constexpr int N = KTL * 4;
    const f32x4* g;
    SmallFrags<KTL> f;
    template <int I> __device__ __forceinline__ void issue() { f.w[I] = g[I * 64]; }
    __device__ __forceinline__ void issue_all() {
#pragma unroll
        for (int i = 0; i < N; ++i) f.w[i] = g[i * 64];
    }
};

// One GEMM stage out[nt] = init(nt) + W_nt^T in, NT output tiles, KT input tiles, spread over the 4 waves.
// Outputs are written to LDS as NT*KS (partial) tiles, index nt*KS + ks; the caller's `in(kt)` functor returns
// input tile kt in accumulator layout (it gathers / activates / gates whatever the producer left in LDS).
template <int NT_, int KT_, int FORCE_KS = 0>
struct SmallStage {
    static constexpr int NT = NT_, KT = KT_;
    using Split = SmallSplit<NT_, KT_, FORCE_KS>;
    static constexpr int KS = Split::KS, KTL = Split::KTL, UNITS = Split::UNITS;
    static constexpr int OUT_TILES = NT_ * Split::KS;

    // issue the global loads of this wave's weight fragments (panels: n-tile major, KT k-tiles each)
    __device__ static __forceinline__ SmallFrags<KTL> fetch(const float* gpanels, int wave, int lane) {
        SmallFrags<KTL> f;
        // NOT under `if (wave < UNITS)`: a load inside a branch makes every later s_waitcnt vmcnt conservative
        // (the compiler must assume either path), which serialises each stage behind the fetch it has just
        // issued for the next one.  Idle waves fetch unit 0's slice and never use it.
        const int u = wave < UNITS ? wave : 0;
        const int nt = u / KS, ks = u % KS;
        const f32x4* g = reinterpret_cast<const f32x4*>(gpanels + ((size_t)nt * KT + ks * KTL) * LSNF_FRAG_FLOATS) + lane;
#pragma unroll
        for (int i = 0; i < KTL * 4; ++i) f.w[i] = g[i * 64];
        return f;
    }

    // start a fetch of this stage's fragments that a running stage will issue (see SmallFetch)
    __device__ static __forceinline__ SmallFetch<KTL> begin_fetch(const float* gpanels, int wave, int lane) {
        SmallFetch<KTL> p;
        const int u = wave < UNITS ? wave : 0;           // idle waves fetch unit 0's slice: no loads under a branch
        const int nt = u / KS, ks = u % KS;
        p.g = reinterpret_cast<const f32x4*>(gpanels + ((size_t)nt * KT + ks * KTL) * LSNF_FRAG_FLOATS) + lane;
        return p;
    }

    // run this stage and, between its MFMA groups, issue the loads of `nf` (the fragments of a later stage)
    template <class InFn, class InitFn, int NKTL>
    __device__ static __forceinline__ void run(const SmallFrags<KTL>& fr, float* out_lds, int wave, int lane, InFn&& in,
                                               InitFn&& init, SmallFetch<NKTL>& nf) {
        if constexpr (UNITS < 4) {
            if (wave >= UNITS) { nf.issue_all(); return; }   // wave-uniform; same number of loads on both paths
        }
        constexpr int G = KTL * 4, L = NKTL * 4, PER = (L + G - 1) / G;
        const int nt = wave / KS, ks = wave % KS;
        f32x16 acc = (ks == 0) ? init(nt) : lsnf_zero16();
        f32x16 x;
        lsnf_static_for<G>([&](auto gi) {
            constexpr int k = gi / 4, g = gi % 4;
            if constexpr (g == 0) {
                if constexpr (std::is_invocable_v<InFn, int, int>) x = in(ks * KTL + k, k);
                else x = in(ks * KTL + k);
            }
            const f32x4 w = fr.w[k * 4 + g];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[4 * g + 3], acc, 0, 0, 0);
            constexpr int L0 = gi * PER, L1 = (L0 + PER < L) ? L0 + PER : L;
            if constexpr (L0 < L) {
                lsnf_static_for<L1 - L0>([&](auto j) { nf.template issue<L0 + j>(); });
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);           // 4 MFMA, then
                __builtin_amdgcn_sched_group_barrier(0x020, L1 - L0, 0);     // this group's VMEM reads
            }
        });
        small_store_tile(out_lds + (size_t)(nt * KS + ks) * LSNF_TILE_FLOATS, acc, lane);
    }

    template <class InFn, class InitFn>
    __device__ static __forceinline__ void run(const SmallFrags<KTL>& fr, float* out_lds, int wave, int lane, InFn&& in,
                                               InitFn&& init) {
        if (wave >= UNITS) return;                       // wave-uniform
        const int nt = wave / KS, ks = wave % KS;
        f32x16 acc = (ks == 0) ? init(nt) : lsnf_zero16();
#pragma unroll
        for (int k = 0; k < KTL; ++k) {
            f32x16 x;   // in(kt) or in(kt, k): k is the compile-time index within this wave's K slice
            if constexpr (std::is_invocable_v<InFn, int, int>) x = in(ks * KTL + k, k);
            else x = in(ks * KTL + k);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = fr.w[k * 4 + g];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[4 * g + 3], acc, 0, 0, 0);
            }
        }
        small_store_tile(out_lds + (size_t)(nt * KS + ks) * LSNF_TILE_FLOATS, acc, lane);
    }
};
