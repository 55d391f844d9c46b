// lsnf_fwd3p.hip -- the bf16x3 throughput forward (lsnf_fwd3.hip: fp32-accurate GEMMs as six bf16 MFMAs per product)
// with its vector work SOFTWARE-PIPELINED under the matrix instructions.
//
// Why.  lsnf_fwd3.hip runs "split the inputs -> MFMA the stage -> ReLU / coupling -> split ..." as separate phases, and its
// 6 270 VALU instructions per wave (operand split, sigmoid / log epilogue) ADD to the 1 920 MFMAs: MFMA-VALU co-execution is
// 4.7 % of MFMA-busy cycles there (PMC).  Two hardware facts decide what can be done about it (tools/micro/shadow.hip,
// tools/micro/stagger.hip): (1) an MFMA of either shape holds the SIMD's VALU issue for ~12 cycles -- the
// 32-cycle 32x32x16 then leaves five 4-cycle slots, the 16-cycle 16x16x32 one; (2) packed fp32 math (v_pk_add_f32 ...) waits
// for the matrix pipe, so this file is compiled without it (Makefile: -fno-slp-vectorize).  The first kernel below is the
// 32x32x16 kernel with every piece of vector work placed, by hand, between the MFMAs of a stage that does not depend on it
// (it needs the fewest cycles, but the chip sustains a ~12 % lower clock under that MFMA shape on real data: opt-in
// LSNF_MATH_BF16X3_PIPE, the reference point); the second, lsnf_fwd3q_kernel, is the same pipeline on 16x16x32 and the
// library's default throughput forward:
//
//   phase (one LDS weight buffer)   MFMAs   vector work carried between them
//   S1a  v[0,1]  += Wa^T x           96     block 0: split x[1..3] one k-tile ahead; later blocks (k order 2,0,1,3):
//                                            split x[0], x[1]  |  sigmoid(p1)  |  x[3] = (v2[1] + t1) * sig, split x[3]
//   S1b  v[2,3]  += Wa^T x           96     split v[0], v[1] (S2's input); last block: store v[0], v[1]
//   S2   h1      += W1'^T v1         48     ReLU + split h1[0] once its last k-step is in
//   S3   h2      += W2'^T h1         48     ReLU + split h1[1]  |  ReLU + split h2[0]
//   S4A  p0, t0  += W3^T h2          48     ReLU + split h2[1]
//   S4B  p1, t1  += W3^T h2          48     sigmoid(p0)  |  x[2] = (v2[0] + t0) * sig, split x[2]
//
// (the loop is rotated: S1a of block b+1 closes the iteration of block b).  Stages run k-major over their two n-tiles so
// that the split of input k-tile 1 has four MFMA groups to hide under.  The interleave is pinned with
// sched_group_barrier (1 MFMA, V VALU); V <= 6 keeps the issue cost inside the MFMA's 32 cycles.
//
// Same math, same prepared weights (plan region off_f3_panels) and the same I/O contract as lsnf_fwd3_kernel; replaces
// reference model.py:473-483 + train.py:317-319.  Covers HT = 2 (nz in 66..128).  The STASH instantiation of lsnf_fwd3q_kernel also
// writes the backward's stash (block outputs, sigma tiles, ReLU mask words: bit for bit what lsnf_fwd3.hip writes) from inside the
// phases -- buffer stores late in each phase, left in flight across the phase barrier by a counted s_waitcnt (stash helpers below);
// calls with the parameter-gradient dump stay on lsnf_fwd3.hip.
#include <stdlib.h>
#include <type_traits>
#include "lsnf_l16.h"

#if LSNF_L16_PARTS != 3
#error "lsnf_fwd3p.hip is the three-term bf16 kernel"
#endif

#ifndef LSNF_FILLMASK
#define LSNF_FILLMASK 0x7f   // diagnostic builds: which phases carry their vector work (register-pressure bisection)
#endif

namespace {

struct Fwd3pArgs {
    const float* consts; const float* panels3;
    const float* z_in; const float* objective;
    float* z_out; float* logdet_out; float* ll_out;
    int B, nz, half, n_blocks, vec4;
    double* stats;
    float* hdump; int width;           // STASH = 2: h1 / h2 of every block into the parameter-gradient dump, TILED form (lsnf_l16.h l16_store_tiled)
    float* z_saved; float* act_saved;  // lsnf_fwd3q_kernel<.., STASH >= 1>: block outputs 0..n_blocks-2 and the activation stash (lsnf_layout.h LsnfActLayout)
    unsigned long long* stamps;        // LSNF_STAMPS diagnostic build only: [waves & 2047][64] clock stamps
};

#ifdef LSNF_STAMPS   // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz; per-phase cycles (tools/stamps_fwd3p.py)
#define P_STAMP(i, INSN)                                                                                \
    do { __builtin_amdgcn_sched_barrier(0);                                                             \
         unsigned long long t_; asm volatile(INSN " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");  \
         __builtin_amdgcn_sched_barrier(0);                                                             \
         if (a.stamps && lane == 0) a.stamps[(((size_t)blockIdx.x * NWAVES + wave) & 2047) * 64 + (i)] = t_; } while (0)
#else
#define P_STAMP(i, INSN) do {} while (0)
#endif

template <int WT_>
struct Fwd3pCfg : LsnfStackCfg<2, WT_> {
    using S = LsnfStackCfg<2, WT_>;
    static constexpr int F = LSNF_FRAG3_FLOATS;
    static constexpr int OFF3_S2 = F * S::P1 * S::KT1;
    static constexpr int OFF3_S3 = OFF3_S2 + F * S::P2 * S::KT2;
    static constexpr int OFF3_S4 = OFF3_S3 + F * S::P3 * S::KT3;
    static constexpr int BLOCK3 = OFF3_S4 + F * S::P4 * S::KT4;
    static constexpr int SLOT3 = 2 * S::MAXKT * F;            // LDS floats of one two-tile buffer
    static constexpr int CONST_FLOATS = S::FWD_CONST;
};

// Vector work whose results are consumed only by a LATER phase would be sunk out of the MFMA region it is meant to hide
// under (instruction selection orders pure arithmetic by use, not by source position; sched_barrier constrains the machine
// scheduler only): an empty volatile asm that reads the results pins them to the step that computed them.
__device__ __forceinline__ void keep(unsigned a, unsigned b, unsigned c) { asm volatile("" :: "v"(a), "v"(b), "v"(c)); }
__device__ __forceinline__ void keep(float a, float b, float c, float d, float e) { asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e)); }

// ---- one MFMA group: acc += A(frag) * x(k-step) for the six kept terms, with V VALU pinned behind every MFMA ----------
struct StepDesc { int acc, tile, s, frag; };     // accumulator index, input tile, k-step, fragment offset (bf16x8 units / 64)
template <int KT> constexpr StepDesc mkstep(int acc, int tl, int in_tile, int kt, int s) { return StepDesc{acc, in_tile, s, ((tl * KT + kt) * 2 + s) * 3}; }

#ifdef LSNF_EXPERIMENTAL_KERNELS   // the 32x32x16 form of the pipeline (research builds: LSNF_MATH value 4), measured slower than the 16x16x32 form
// the three bf16 terms of one activation tile, B-operand order: w[s][part] = k-step s (registers 8s..8s+7)
struct SplitTile { unsigned d[2][3][4]; };    // scalars, assembled into the 128-bit operand at the MFMA: a partial write
                                               // of a register TUPLE keeps the whole old tuple alive (measured: +170 VGPRs)

// registers 4q..4q+3 of `x` (optionally through ReLU) -> dwords 2(q&1), 2(q&1)+1 of k-step q>>1: 18 (22) VALU
template <bool RELU>
__device__ __forceinline__ void split_quad(const f32x16& x, int q, SplitTile& out) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float a = x[4 * q + 2 * j], b = x[4 * q + 2 * j + 1];
        if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
        const unsigned p1 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p1 << 16); b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned p2 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        const int d = 2 * (q & 1) + j;
        out.d[q >> 1][0][d] = p1; out.d[q >> 1][1][d] = p2; out.d[q >> 1][2][d] = pk_bf16(a, b);
        keep(out.d[q >> 1][0][d], out.d[q >> 1][1][d], out.d[q >> 1][2][d]);
    }
}
template <bool RELU>
__device__ __forceinline__ void split_tile(const f32x16& x, SplitTile& out) {
#pragma unroll
    for (int q = 0; q < 4; ++q) split_quad<RELU>(x, q, out);
}

// sigmoid / log2 of registers 4q..4q+3: p <- sigmoid(p), lsum += log2(1 + exp(-p))   (model.py:413,418)
__device__ __forceinline__ void sigmoid_quad(f32x16& p, int q, float& lsum) {
#pragma unroll
    for (int r = 4 * q; r < 4 * q + 4; ++r) {
        float sig, l2;
        lsnf_sigmoid_log2(p[r], sig, l2);
        p[r] = sig; lsum += l2;
    }
    keep(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3], lsum);
}
// coupling on registers 4q..4q+3: v <- (v + t) * sig   (model.py:414-415)
__device__ __forceinline__ void couple_quad(f32x16& v, const f32x16& t, const f32x16& sig, int q) {
#pragma unroll
    for (int r = 4 * q; r < 4 * q + 4; ++r) v[r] = (v[r] + t[r]) * sig[r];
}

template <int V>
__device__ __forceinline__ void pin6() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (V > 0) __builtin_amdgcn_sched_group_barrier(0x402, V, 0);   // VALU | transcendental
    }
}

// PH: struct with  static constexpr int N, MID;  static constexpr StepDesc at(int i);  static constexpr int valu(int i);
// fill(ic): vector work of step ic (generic lambda on std::integral_constant); mid(): called between steps MID-1 and MID
template <class PH, class Fill, class Mid>
__device__ __forceinline__ void run_phase(f32x16* acc, const SplitTile* in, const float* lbuf, int lane, Fill&& fill, Mid&& mid) {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(lbuf) + lane;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = wp[(PH::at(0).frag + p) * 64];
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    lsnf_static_for<PH::N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr StepDesc d = PH::at(i);
        if constexpr (i == PH::MID) { mid(); __builtin_amdgcn_sched_barrier(0); }
        bf16x8 na[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) na[p] = a[p];
        if constexpr (i + 1 < PH::N) {
#pragma unroll
            for (int p = 0; p < 3; ++p) na[p] = wp[(PH::at(i + 1).frag + p) * 64];
        }
#ifndef LSNF_ABL_NOFILL     // timing diagnostics only (wrong numbers): tools/ablate_fwd3p.sh
        fill(ic);
#endif
        const SplitTile& x = in[d.tile];
#define LSNF_P_MMA(WI, XI) \
        acc[d.acc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[WI], __builtin_bit_cast(bf16x8, u32x4{x.d[d.s][XI][0], x.d[d.s][XI][1], x.d[d.s][XI][2], x.d[d.s][XI][3]}), acc[d.acc], 0, 0, 0);
#ifdef LSNF_ABL_NOMFMA
        LSNF_P_MMA(0, 0)
#else
        LSNF_F3_TERMS(LSNF_P_MMA)
#endif
#undef LSNF_P_MMA
        if constexpr (i + 1 < PH::N) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        pin6<PH::valu(i)>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = na[p];
    });
}

// ---- phase tables: every phase is 16 MFMA groups (96 MFMAs) out of one 48 KiB LDS buffer --------------------------
// S1a: v[0], v[1] (accumulators 0,1 = buffer tiles 0,1), k order 2,3,0,1: x[2]'s split is ready; x[3] is split under
// k-tile 2, x[0] under 3, x[1] under 0
struct PhS1a {
    static constexpr int N = 16, MID = 8;
    static constexpr int korder(int j) { return j == 0 ? 2 : (j == 1 ? 3 : (j == 2 ? 0 : 1)); }
    static constexpr StepDesc at(int i) { return mkstep<4>((i >> 1) & 1, (i >> 1) & 1, korder(i >> 2), korder(i >> 2), i & 1); }
    static constexpr int valu(int i) { return i < 12 ? 3 : 0; }
};
// S1b: v[2], v[3] (array offset by the caller), all inputs ready; v[0], v[1] are split under the first eight groups
struct PhS1b {
    static constexpr int N = 16, MID = 8;
    static constexpr StepDesc at(int i) { return mkstep<4>((i >> 1) & 1, (i >> 1) & 1, i >> 2, i >> 2, i & 1); }
    static constexpr int valu(int i) { return i < 8 ? 3 : 0; }
};
// S2 + S3 out of one buffer [h1_0][h1_1][h2_0][h2_1] (two k-tiles each).  Accumulators 0,1 = h1, 2,3 = h2; inputs 0,1 =
// split v[0..1], 2,3 = split h1[0..1].  Both k-major: groups 0-7 S2 (h1[0] complete after group 5), 8-15 S3.
struct PhS23 {
    static constexpr int N = 16, MID = 8;
    static constexpr StepDesc at(int i) {
        const int st = i >> 3, j = i & 7, tl = (j >> 1) & 1, kt = j >> 2;
        return mkstep<2>(2 * st + tl, 2 * st + tl, 2 * st + kt, kt, j & 1);
    }
    //                                  S2: relu+split h1[0] under 6,7 | S3: h1[1] under 8-11, h2[0] under 14,15
    static constexpr int valu(int i) { return (i == 6 || i == 7 || i == 14 || i == 15) ? 8 : ((i >= 8 && i < 12) ? 4 : 0); }
};
// S4 out of one buffer [t0][t1][p0][p1] (fc_zeros panels, two k-tiles each): n-major p0, t0, p1, t1 = accumulators 0..3
struct PhS4 {
    static constexpr int N = 16, MID = 8;
    static constexpr StepDesc at(int i) {
        const int q = i >> 2, tl = (q == 0) ? 2 : (q == 1 ? 0 : (q == 2 ? 3 : 1)), kt = (i >> 1) & 1;
        return mkstep<2>(q, tl, kt, kt, i & 1);
    }
    //          relu+split h2[1] under 0,1 | sigmoid(p0) under 4-7 | x[2] coupling + split under 8-11 | sigmoid(p1) under 12-15
    static constexpr int valu(int i) { return i < 2 ? 8 : (i < 4 ? 0 : (i < 8 ? 6 : (i < 12 ? 5 : 6))); }
};

template <int WT, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES, 1) void lsnf_fwd3p_kernel(const Fwd3pArgs a) {
    using C = Fwd3pCfg<WT>;
    static_assert(WT == 2, "lsnf_fwd3p_kernel: f_width <= 64 instantiation");
    constexpr int THREADS = 64 * NWAVES;
    constexpr int HT = 2, NZT = 4, F = C::F;
    constexpr int SLOT = C::SLOT3;                             // 48 KiB
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;                                         // n_blocks * CONST_FLOATS
    float* const buf0 = smem + a.n_blocks * C::CONST_FLOATS;   // 3 x SLOT
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 31, h = lane >> 5;
    // Two waves share a SIMD (wave w and w + 4).  The second half of the workgroup runs HALF A PHASE behind the first:
    // it meets the workgroup barrier of phase k in the middle of its phase k-1.  What one wave spends outside its MFMA
    // stream at a phase boundary (barrier skew, LDS-DMA issue, first fragment reads, the few exposed VALU: ~1 800 cycles,
    // tools/stamps_fwd3p.py) then falls beside its partner's MFMAs instead of beside its partner's identical pause.
    // Needs a third weight buffer: at barrier k the late half still reads buffer k-1 while buffer k+1 is being filled.
#ifdef LSNF_ABL_NOSTAGGER
    const bool late = false;
#else
    const bool late = NWAVES == 8 && wave >= 4;                // wave-uniform
#endif

    P_STAMP(0, "s_memtime");
    P_STAMP(50, "s_memrealtime");
    const int n_phases = 4 * a.n_blocks;
    // phase k of the launch: block k >> 2; (k & 3) = 0 S1a, 1 S1b, 2 S2+S3, 3 S4 -- each ONE contiguous 48 KiB of the stream
    auto phase_src = [&](int k) -> const float* {
        const float* gb = a.panels3 + (size_t)(k >> 2) * C::BLOCK3;
        const int j = k & 3;
        return gb + (j == 0 ? 0 : (j == 1 ? 2 * 4 * F : (j == 2 ? C::OFF3_S2 : C::OFF3_S4)));
    };
    // barrier k: my LDS-DMA has landed (vmcnt), everybody's has (barrier), buffer (k+1) % 3 is free -> start phase k+1's panels
    auto sync_issue = [&](int k) {
#ifndef LSNF_ABL_NOSYNC
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // (reached by the two halves at different points of their instruction streams,
                                               //  the same number of times: wave-uniform control flow around a convergent call)
#endif
#ifndef LSNF_ABL_NODMA
        if (k + 1 < n_phases) issue_kib<48, NWAVES>(phase_src(k + 1), buf0 + ((k + 1) % 3) * SLOT, wave, lane);
#endif
    };
    issue_kib<48, NWAVES>(phase_src(0), buf0, wave, lane);
    for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += THREADS) cst[i] = a.consts[i];

    const long sample = ((long)blockIdx.x * NWAVES + wave) * 32 + m;
    const bool live = sample < a.B;
    const long row = live ? sample : (long)a.B - 1;

    // Loop-carried state: xin[0..1] = first half of the block input (fp32; = v1 of the previous block), v[3] = last tile of
    // the second half (coupled, fp32), xs[2] = split of the third tile, ell.  Every iteration runs the same code:
    // S1a (which finishes the split of its own input under its first MFMAs) .. S4; no branch leaves the loop body.
    f32x16 xin[2];
    f32x16 v[NZT];                // S1 accumulators; v[2..3] become x[2..3] through the coupling
    SplitTile xs[NZT];            // split of the block input
    {
        f32x16 x[NZT];
        lsnf_load_rows<HT>(x, a.z_in, row, a.nz, a.half, h, a.vec4);
        xin[0] = x[0]; xin[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    }
    float ell = a.objective ? a.objective[row] : 0.0f;
    split_tile<false>(v[2], xs[2]);                 // the only split that is not hidden (once per launch)
    float ss01 = 0.0f;                              // sum of squares of the (final) first half, taken when it is stored
    sync_issue(0);                                  // barrier 0: both halves, before their first phase
    P_STAMP(1, "s_memtime");

    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const bool more = blk + 1 < a.n_blocks;
        const int k0 = 4 * blk;
        SplitTile vh[4];                            // S2+S3 inputs: split v[0], v[1], h1[0], h1[1]
        SplitTile h2s[WT];
        f32x16 hh[4], tp[4];                        // hh: h1[0], h1[1], h2[0], h2[1];  tp: p0, t0, p1, t1
        float lsum = 0.0f;
        auto mid_sync = [&](int k) { if (late && k + 1 < n_phases) sync_issue(k + 1); };

        if (blk == 1) P_STAMP(10, "s_memtime");
        // ---- S1a: v[0,1] (k order 2,3,0,1): split x[3] | x[0] | x[1] under k-tiles 2 | 3 | 0  (model.py:187; actnorm :244,268 folded) ----
        {
            v[0] = lsnf_bias_init(cb + 0, h); v[1] = lsnf_bias_init(cb + 32, h);
            if (!late && k0 > 0) sync_issue(k0);
            run_phase<PhS1a>(v, xs, buf0 + (k0 % 3) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 4) split_quad<false>(v[3], i, xs[3]);
                else if constexpr (i < 12) split_quad<false>(xin[(i - 4) >> 2], i & 3, xs[(i - 4) >> 2]);
            }, [&] { mid_sync(k0); });
        }
        if (blk == 1) P_STAMP(11, "s_memtime");
        // ---- S1b: v[2,3]; split v[0], v[1] for S2 under the first groups ----
        {
            v[2] = lsnf_bias_init(cb + 64, h); v[3] = lsnf_bias_init(cb + 96, h);
            if (!late) sync_issue(k0 + 1);
            run_phase<PhS1b>(v + 2, xs, buf0 + ((k0 + 1) % 3) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 8) split_quad<false>(v[i >> 2], i & 3, vh[i >> 2]);
            }, [&] { mid_sync(k0 + 1); });
        }
        if (blk == 1) P_STAMP(12, "s_memtime");
        if (!more) {             // last block: the v1 half is final (model.py:422) -- its stores drain under S2..S4
            if (live) {
                float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
                for (int t = 0; t < HT; ++t) lsnf_store_tile<HT>(t, v[t], zo, a.half, h, a.vec4);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ss01 += v[t][r] * v[t][r];
        }
        ell = ell + cb[32 * C::NP + 0];          // sum(3*logs)  (model.py:273-276)
        ell = ell + cb[32 * C::NP + 1];          // log|det W|   (model.py:182,189)
        // ---- S2 + S3: h1 = relu(actnorm(v1 @ W1)), h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,307-308) ----
        {
#pragma unroll
            for (int t = 0; t < 4; ++t) hh[t] = lsnf_bias_init(cb + 32 * (C::P1 + t), h);
            if (!late) sync_issue(k0 + 2);
            run_phase<PhS23>(hh, vh, buf0 + ((k0 + 2) % 3) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i == 6 || i == 7) { split_quad<true>(hh[0], 2 * (i - 6), vh[2]); split_quad<true>(hh[0], 2 * (i - 6) + 1, vh[2]); }
                if constexpr (i >= 8 && i < 12) split_quad<true>(hh[1], i - 8, vh[3]);
                if constexpr (i >= 14) { split_quad<true>(hh[2], 2 * (i - 14), h2s[0]); split_quad<true>(hh[2], 2 * (i - 14) + 1, h2s[0]); }
            }, [&] { mid_sync(k0 + 2); });
        }
        if (blk == 1) P_STAMP(13, "s_memtime");
        xin[0] = v[0]; xin[1] = v[1];            // v1 is the next block's first half (model.py:422)
        // ---- S4: p0, t0, p1, t1 = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) + coupling (:414-418) ----
        {
            constexpr int B4 = C::P1 + C::P2 + C::P3;
            tp[0] = lsnf_bias_init(cb + 32 * (B4 + HT), h); tp[1] = lsnf_bias_init(cb + 32 * (B4 + 0), h);
            tp[2] = lsnf_bias_init(cb + 32 * (B4 + HT + 1), h); tp[3] = lsnf_bias_init(cb + 32 * (B4 + 1), h);
            if (!late) sync_issue(k0 + 3);
            run_phase<PhS4>(tp, h2s, buf0 + ((k0 + 3) % 3) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 2) { split_quad<true>(hh[3], 2 * i, h2s[1]); split_quad<true>(hh[3], 2 * i + 1, h2s[1]); }
                if constexpr (i >= 4 && i < 8) sigmoid_quad(tp[0], i - 4, lsum);
                if constexpr (i >= 8 && i < 12) { couple_quad(v[2], tp[1], tp[0], i - 8); split_quad<false>(v[2], i - 8, xs[2]); }   // (split unused after the last block)
                if constexpr (i >= 12) sigmoid_quad(tp[2], i - 12, lsum);
            }, [&] { mid_sync(k0 + 3); });
#pragma unroll
            for (int q = 0; q < 4; ++q) couple_quad(v[3], tp[3], tp[2], q);   // x[3] (32 VALU, not hidden)
        }
        ell = ell + -0.6931471805599453f * lsnf_pair_sum(lsum);
        if (blk == 1) P_STAMP(14, "s_memtime");
    }
    P_STAMP(40, "s_memtime");

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
    float ss = ss01;
#pragma unroll
    for (int t = HT; t < NZT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += v[t][r] * v[t][r];
    ss = lsnf_pair_sum(ss);
    if (live) {
        float* zo = a.z_out + sample * (long)a.nz;
#pragma unroll
        for (int t = HT; t < NZT; ++t) lsnf_store_tile<HT>(t, v[t], zo, a.half, h, a.vec4);
        if (h == 0) {
            a.logdet_out[sample] = ell;
            if (a.ll_out) a.ll_out[sample] = (-0.5f * ss + 1.8378770664093453f) + ell;
        }
    }
    if (a.stats) {   // kernel-uniform: batch sums of ll and logdet, one pair of fp64 atomics per workgroup
        float sl = (live && h == 0) ? ((-0.5f * ss + 1.8378770664093453f) + ell) : 0.0f;
        float sd = (live && h == 0) ? ell : 0.0f;
        double dl = (double)sl, dd = (double)sd;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
        __syncthreads();
        double* red = reinterpret_cast<double*>(buf0);
        if (lane == 0) { red[2 * wave] = dl; red[2 * wave + 1] = dd; }
        __syncthreads();
        if (wave == 0) {         // (all 64 lanes: lsnf_publish_stats is a wave-level protocol)
            double tl = 0.0, td = 0.0;
            for (int w = 0; w < NWAVES; ++w) { tl += red[2 * w]; td += red[2 * w + 1]; }
            lsnf_publish_stats(a.stats, tl, td, a.B, lane);
        }
    }
    P_STAMP(41, "s_memtime");
    P_STAMP(51, "s_memrealtime");
}

#endif  // LSNF_EXPERIMENTAL_KERNELS

// =====================================================================================================================
// The same pipeline on v_mfma_f32_16x16x32_bf16 (the "L16" lane layout of lsnf_l16.h, the weights of plan region
// off_f3b_panels).  Why a second form: the 32x32x16 kernel above needs fewer cycles than lsnf_fwd3b_kernel but the chip
// sustains a ~12 % lower clock under that MFMA shape.  tools/micro/shadow.hip: an MFMA of either shape holds the SIMD's
// VALU issue for ~12 cycles; the 16-cycle 16x16x32 then leaves ONE 4-cycle slot, and one independent VALU per MFMA in that
// slot is free (7.41 vs 7.20 ns per MFMA with two waves per SIMD), two cost +15 %.  So here a step = 12 MFMAs (one
// 16-feature half of an output tile x one 32-feature k-tile x two sample tiles x six terms) carries 12..24 VALU.
// =====================================================================================================================
struct Tile16 { f32x4v q[4]; };                  // q[2*ft + st]: features 16*ft + 4*g + r of sample 16*st + n
struct SplitTile16 { unsigned d[2][3][4]; };     // [st][part][dword 2*ft + j] = the B operands of one k-tile

// ST = sample tiles per wave: 2 (32 rows per wave) or 1 (16 rows per wave: the same kernel for launches that would otherwise
// leave SIMDs idle -- twice the workgroups per row; the quads q = 2*ft + 1 of a Tile16 and d[1] of a SplitTile16 are unused)
template <int HT, int ST>
__device__ __forceinline__ Tile16 load_tile16(int t, const float* __restrict__ z, const int* rows, int nz, int half, int g, int vw) {
    Tile16 x;
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int f0 = 32 * tt + l16_feat0(ft, g), col0 = hh * half + f0, q = 2 * ft + st;
            const float* zr = z + (long)rows[st] * (long)nz;
            if (vw == 4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (f0 < half) v = *reinterpret_cast<const f32x4*>(zr + col0);
                x.q[q][0] = v[0]; x.q[q][1] = v[1]; x.q[q][2] = v[2]; x.q[q][3] = v[3];
            } else if (vw == 2) {
                f32x2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
                if (f0 < half) v0 = *reinterpret_cast<const f32x2*>(zr + col0);
                if (f0 + 2 < half) v1 = *reinterpret_cast<const f32x2*>(zr + col0 + 2);
                x.q[q][0] = v0[0]; x.q[q][1] = v0[1]; x.q[q][2] = v1[0]; x.q[q][3] = v1[1];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x.q[q][j] = (f0 + j < half) ? zr[col0 + j] : 0.0f;
            }
        }
    return x;
}
template <int HT, int ST>
__device__ __forceinline__ void store_tile16(int t, const Tile16& x, float* __restrict__ z, const int* rows, const bool* live,
                                             int nz, int half, int g, int vw) {
    const int hh = t / HT, tt = t % HT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            if (!live[st]) continue;
            const int f0 = 32 * tt + l16_feat0(ft, g), col0 = hh * half + f0, q = 2 * ft + st;
            float* zr = z + (long)rows[st] * (long)nz;
            if (vw == 4) {
                if (f0 < half) { f32x4 v = {x.q[q][0], x.q[q][1], x.q[q][2], x.q[q][3]}; *reinterpret_cast<f32x4*>(zr + col0) = v; }
            } else if (vw == 2) {
                if (f0 < half) { f32x2 v = {x.q[q][0], x.q[q][1]}; *reinterpret_cast<f32x2*>(zr + col0) = v; }
                if (f0 + 2 < half) { f32x2 v = {x.q[q][2], x.q[q][3]}; *reinterpret_cast<f32x2*>(zr + col0 + 2) = v; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f0 + j < half) zr[col0 + j] = x.q[q][j];
            }
        }
}
// The bias of an output tile (its 2 x 4 values per lane) is the C operand of the FIRST MFMA of each accumulator chain: both
// sample tiles start from the same registers, nothing is copied (an initialised accumulator pair would cost 8 v_mov per tile).
// It is read from the constant block in LDS one step ahead of that MFMA (4 registers in flight instead of 8 per tile up front).
__device__ __forceinline__ const float* bias_lane_ptr(const float* cb, int g) { return cb + (g & 1) * 16 + 4 * (g >> 1); }
__device__ __forceinline__ f32x4v bias_quad(const float* lane_ptr, int float_off) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(lane_ptr + float_off);
    f32x4v r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = v[j];
    return r;
}
template <class PH> constexpr bool first_touch(int i) {       // is step i the first one on its (accumulator, ft)?
    for (int j = 0; j < i; ++j)
        if (PH::at(j).acc == PH::at(i).acc && PH::at(j).s == PH::at(i).s) return false;
    return true;
}
// One unit of vector work, indexed q = 0..3 in the phase tables.  ST = 2: quad q = 2*ft + st of a tile (4 values per lane).
// ST = 1: the tile has the quads 2*ft only, and unit q = 2*ft + j is the element PAIR j of quad 2*ft -- half the work per
// unit, under half the MFMAs per step, with the same readiness (a unit depends on the 16-feature half ft = q >> 1 only).
template <int ST> __device__ __forceinline__ constexpr int unit_quad(int q) { return ST == 2 ? q : (q & ~1); }
template <int ST> __device__ __forceinline__ constexpr int unit_pair0(int q) { return ST == 2 ? 0 : (q & 1); }
template <int ST> __device__ __forceinline__ constexpr int unit_st(int q) { return ST == 2 ? (q & 1) : 0; }
// unit q (optionally through ReLU) -> its dwords of the B operands: 22 (26) VALU per quad
template <bool RELU, int ST>
__device__ __forceinline__ void split_q16(const Tile16& x, int q, SplitTile16& out) {
    // (no contraction: behind the coupling, "x - p1" would become fma(v2 + t, sigma, -p1) and the terms would describe the unrounded
    //  product instead of the fp32 value that every other kernel -- and the block's stored output -- uses: rows would then depend,
    //  in their last bits, on which kernel the batch size selects)
#pragma clang fp contract(off)
    const int ft = q >> 1, st = unit_st<ST>(q), qr = unit_quad<ST>(q);
#pragma unroll
    for (int jj = 0; jj < ST; ++jj) {
        const int j = unit_pair0<ST>(q) + jj;
        float a = x.q[qr][2 * j], b = x.q[qr][2 * j + 1];
        if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
        const unsigned p1 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p1 << 16); b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned p2 = pk_bf16(a, b);
        a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        const int d = 2 * ft + j;
        out.d[st][0][d] = p1; out.d[st][1][d] = p2; out.d[st][2][d] = pk_bf16(a, b);
        keep(out.d[st][0][d], out.d[st][1][d], out.d[st][2][d]);
    }
}
template <int ST>
__device__ __forceinline__ void sigmoid_q16(Tile16& p, int q, float* lsum /*[2]: per sample tile*/) {
    const int st = unit_st<ST>(q), qr = unit_quad<ST>(q), r0 = 2 * unit_pair0<ST>(q);
#pragma unroll
    for (int r = r0; r < r0 + 2 * ST; ++r) {
        float sig, l2;
        lsnf_sigmoid_log2(p.q[qr][r], sig, l2);
        p.q[qr][r] = sig; lsum[st] += l2;
    }
    keep(p.q[qr][r0], p.q[qr][r0 + 1], p.q[qr][r0 + 2 * ST - 2], p.q[qr][r0 + 2 * ST - 1], lsum[st]);
}
template <int ST>
__device__ __forceinline__ void couple_q16(Tile16& v, const Tile16& t, const Tile16& sig, int q) {
    const int qr = unit_quad<ST>(q), r0 = 2 * unit_pair0<ST>(q);
#pragma unroll
    for (int r = r0; r < r0 + 2 * ST; ++r) v.q[qr][r] = (v.q[qr][r] + t.q[qr][r]) * sig.q[qr][r];
}

// ---- stash writes of the pipelined kernel (STASH instantiation; ST = 2: a wave's 32 rows are one stash tile) -------------------------
// All of them are raw BUFFER stores: one straight-line instruction each -- no branch splits the pinned schedule of a step --, and the
// descriptor's byte count drops what must not land: rows past the batch, the tiles of a wave past the batch, and (offset 2^31) the
// 4-feature groups past `half` of a row's padded second tile.
#ifndef LSNF_STASH_AUX
#define LSNF_STASH_AUX 0      // cache-policy bits of the stash stores (experiment knob: 2 = nt)
#endif
#ifndef LSNF_STASH_PARTS
#define LSNF_STASH_PARTS 7    // timing diagnostics (wrong stash): 1 = ReLU masks, 2 = sigma tiles, 4 = block output rows
#endif
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
struct StashLane { unsigned zoff, soff, moff, msh, hoff; };   // per-lane byte offsets: row store, sigma tile, mask words; mask shift 4 * (g >> 1)
// (every store below has soffset = 0: a uniform byte offset -- the second half of a row, the second sigma tile -- is a second descriptor,
//  base + off / bytes - off, formed on the scalar unit.  With a REGISTER soffset the compiler's hazard recognizer takes a 128-bit
//  buffer store for safe against an immediately following VALU write of its data registers; on this chip it is not: the first data
//  dword of a few lanes then carried the NEW value -- found as 52 wrong rows out of 65 536 in this kernel's first version)
struct StashRsrc { __amdgpu_buffer_rsrc_t r[2]; };
__device__ __forceinline__ StashRsrc stash_rsrc(const float* p, unsigned bytes, unsigned second_off) {
    char* c = reinterpret_cast<char*>(const_cast<float*>(p));
    return StashRsrc{{__builtin_amdgcn_make_buffer_rsrc(c, 0, (int)bytes, 0x00020000),
                      __builtin_amdgcn_make_buffer_rsrc(c + second_off, 0, (int)(bytes > second_off ? bytes - second_off : 0u), 0x00020000)}};
}
// quad q = 2*ft + st of tile t -> its 16 bytes of row `16*st + n` (the layout of store_tile16 at vec4 == 4)
template <int HT>
__device__ __forceinline__ void stash_row_quad(const StashRsrc& rs, const StashLane& sl, int t, int q, const Tile16& x, int nz, int half, int g) {
    if constexpr (!(LSNF_STASH_PARTS & 4)) return;
    const int ft = q >> 1, st = q & 1, hh = t / HT, tt = t % HT;
    unsigned off = sl.zoff;
    if (st) off += (unsigned)(16 * nz * 4);
    off = (32 * tt + 16 * ft + 4 * g < half) ? off : 0x80000000u;        // (nz <= 64 runs on this HT = 2 kernel too: half < 32 pads tile 0 as well)
    // (the whole vector is cast: __builtin_bit_cast of ONE element of an ext_vector lvalue reads element 0 with this compiler)
    const u32x4s v = __builtin_bit_cast(u32x4s, x.q[q]);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs.r[hh], off + (unsigned)((32 * tt + 16 * ft) * 4), 0, LSNF_STASH_AUX);
}
// sigma quad q of feature tile t (l16_store_sigma's layout: [2*ft + (g >> 1)][16*st + n + 32*(g & 1)][4])
__device__ __forceinline__ void stash_sigma_quad(const StashRsrc& rs, const StashLane& sl, int t, int q, const Tile16& sg) {
    if constexpr (!(LSNF_STASH_PARTS & 2)) return;
    const int ft = q >> 1, st = q & 1;
    const u32x4s v = __builtin_bit_cast(u32x4s, sg.q[q]);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs.r[t], sl.soff + (unsigned)((2 * ft * 64 + 16 * st) * 16), 0, LSNF_STASH_AUX);
}
// ReLU bits of quad q of a hidden tile: bit r = (h[r] > 0).  The sign of (+0 - h) IS that predicate (-(+-0) = +0, no NaNs here), and
// v_alignbit shifts it in: 2 VALU per value instead of compare + select + shift/or.
__device__ __forceinline__ void mask_q16(const Tile16& h, int q, unsigned (&mk)[2][2]) {
    if constexpr (!(LSNF_STASH_PARTS & 1)) return;
    unsigned m = 0;
#pragma unroll
    for (int r = 3; r >= 0; --r) m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned, 0.0f - h.q[q][r]), 31);
    mk[q & 1][q >> 1] = m;
    asm volatile("" :: "v"(m));
}
// the two mask words of a tile (l16_store_masks' format: bit 4*(2*ft + (g >> 1)) + r of word 16*st + n + 32*(g & 1)); all 64 lanes
// store -- lanes l and l + 32 hold the same word after the exchange and write it to the same address
__device__ __forceinline__ void stash_mask_words(const StashRsrc& rs, const StashLane& sl, int tile, const unsigned (&mk)[2][2]) {
    if constexpr (!(LSNF_STASH_PARTS & 1)) return;
    const unsigned w0 = mk[0][0] | (mk[0][1] << 8), w1 = mk[1][0] | (mk[1][1] << 8);
    unsigned x = ((w0 << 16) | w1) << sl.msh;
    x = lsnf_pair_or32(x);
    __builtin_amdgcn_raw_buffer_store_b32(x >> 16, rs.r[0], sl.moff + (unsigned)(tile * 64 * 4), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(x & 0xffffu, rs.r[0], sl.moff + (unsigned)((tile * 64 + 16) * 4), 0, 0);
}
// quad q = 2*ft + st of hidden tile t, through the ReLU, into the tiled h array of the parameter-gradient dump (l16_store_tiled's
// order: unit ((2t + ft) * 2 + st) * 64 + n * 4 + g of the wave's 32-sample tile; rs.r[t] starts at feature tile t; a padded half
// of the last feature tile gets the out-of-range offset)
__device__ __forceinline__ void stash_h_quad(const StashRsrc& rs, const StashLane& sl, int t, int q, const Tile16& h, int width) {
    const int ft = q >> 1, st = q & 1;
    f32x4v r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = fmaxf(h.q[q][j], 0.0f);         // (the same v_max as the split's: one instruction serves both)
    const unsigned off = (32 * t + 16 * ft < width) ? sl.hoff : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, r), rs.r[t], off + (unsigned)((ft * 2 + st) * 1024), 0, LSNF_STASH_AUX);
}

template <int NM, int NV>
__device__ __forceinline__ void pin_mfma() {       // NM MFMAs, NV VALU spread behind them (the first NV % NM slots carry one more)
    lsnf_static_for<NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value, V = NV / NM + (m < NV % NM ? 1 : 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (V > 0) __builtin_amdgcn_sched_group_barrier(0x402, V, 0);
    });
}
// StepDesc here: acc = output tile, s = its 16-feature half ft, tile = input k-tile, frag as above
template <class PH, int ST, class Fill, class Mid>
__device__ __forceinline__ void run_phase16(Tile16* acc, const float* bias_ptr /* bias_lane_ptr of the block */, const SplitTile16* in, const float* lbuf, int lane, Fill&& fill, Mid&& mid) {
    // mid(ic): called at the head of every step (the kernel spreads the next phase's weight DMA over the steps there)
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(lbuf) + lane;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = wp[(PH::at(0).frag + p) * 64];
    f32x4v bq = bias_quad(bias_ptr, 32 * PH::btile(PH::at(0).acc) + 8 * PH::at(0).s);       // (step 0 is a first touch)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    lsnf_static_for<PH::N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr StepDesc d = PH::at(i);
        mid(ic);
        bf16x8 na[3];
        f32x4v nbq = bq;
#pragma unroll
        for (int p = 0; p < 3; ++p) na[p] = a[p];
        if constexpr (i + 1 < PH::N) {
#pragma unroll
            for (int p = 0; p < 3; ++p) na[p] = wp[(PH::at(i + 1).frag + p) * 64];
            if constexpr (first_touch<PH>(i + 1)) nbq = bias_quad(bias_ptr, 32 * PH::btile(PH::at(i + 1).acc) + 8 * PH::at(i + 1).s);
        }
#ifndef LSNF_ABL_NOFILL
        fill(ic);
#endif
        const SplitTile16& x = in[d.tile];
        constexpr bool first = first_touch<PH>(i);
#define LSNF_Q_MMA(WI, XI) \
        acc[d.acc].q[2 * d.s + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[WI], __builtin_bit_cast(bf16x8, u32x4{x.d[0][XI][0], x.d[0][XI][1], x.d[0][XI][2], x.d[0][XI][3]}), (first && WI == 2 && XI == 0) ? bq : acc[d.acc].q[2 * d.s + 0], 0, 0, 0); \
        if constexpr (ST == 2) acc[d.acc].q[2 * d.s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[WI], __builtin_bit_cast(bf16x8, u32x4{x.d[1][XI][0], x.d[1][XI][1], x.d[1][XI][2], x.d[1][XI][3]}), (first && WI == 2 && XI == 0) ? bq : acc[d.acc].q[2 * d.s + 1], 0, 0, 0);
#ifdef LSNF_ABL_NOMFMA
        LSNF_Q_MMA(0, 0)
#else
        LSNF_F3_TERMS(LSNF_Q_MMA)
#endif
#undef LSNF_Q_MMA
        if constexpr (i + 1 < PH::N) __builtin_amdgcn_sched_group_barrier(0x100, first_touch<PH>(i + 1) ? 4 : 3, 0);
        pin_mfma<6 * ST, (PH::valu(i) * ST) / 2>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = na[p];
        bq = nbq;
    });
}

// ---- phase tables (16 steps = 192 MFMAs out of one 48 KiB buffer); valu(i) = VALU instructions pinned into step i -------------
// VALU per quad (this file is compiled without packed fp32 math, see the Makefile): split 22, ReLU + split 26, sigmoid / log2 28,
// coupling 8.  The loop is rotated: S1a of block b+1 carries the END of block b's coupling.
// S1a: v[0], v[1], natural k order.  Its inputs x[0], x[1] ARE the previous block's v[0], v[1], whose split (S2's input, made
// under S1b) is simply kept -- no second split; x[2]'s split is finished under steps 0-1, then sigmoid(p1) (2-5), the coupling
// of x[3] (6-7) and its split (8-11: k-tile 3 starts at step 12) of the previous block
struct QhNoStores { static constexpr int stores(int) { return 0; } };     // (stash stores per step: the STASH tables below)
struct QhS1a : QhNoStores {
    static constexpr int N = 16;
    static constexpr int btile(int acc) { return acc; }
    static constexpr StepDesc at(int i) { return mkstep<4>((i >> 1) & 1, (i >> 1) & 1, i >> 2, i >> 2, i & 1); }
    static constexpr int valu(int i) { return i < 2 ? 22 : (i < 6 ? 28 : (i < 8 ? 16 : (i < 12 ? 22 : 0))); }
};
// S1b: v[2], v[3]; split v[0], v[1] (S2's input and the next block's x[0], x[1]) under the last eight steps
struct QhS1b : QhNoStores {
    static constexpr int N = 16;
    static constexpr int btile(int acc) { return 2 + acc; }
    static constexpr StepDesc at(int i) { return mkstep<4>((i >> 1) & 1, (i >> 1) & 1, i >> 2, i >> 2, i & 1); }
    static constexpr int valu(int i) { return i >= 8 ? 22 : 0; }
};
// S2 + S3 out of one buffer [h1_0][h1_1][h2_0][h2_1].  S2 n-major (h1[0] is complete after step 3 and is split under h1[1]'s
// steps 4-7), S3 k-major (h1[1] is split under its k-tile-0 steps 8-11; h2's halves complete at steps 12..15)
struct QhS23 : QhNoStores {
    static constexpr int N = 16;
    static constexpr int btile(int acc) { return 4 + acc; }                 // P1 = 4 (nz in 66..128)
    static constexpr StepDesc at(int i) {
        if (i < 8) { const int tl = i >> 2, kt = (i >> 1) & 1; return mkstep<2>(tl, tl, kt, kt, i & 1); }
        const int j = i - 8, kt = j >> 2, tl = (j >> 1) & 1;
        return mkstep<2>(2 + tl, 2 + tl, 2 + kt, kt, j & 1);
    }
    static constexpr int valu(int i) { return (i >= 4 && i < 12) ? 26 : (i >= 13 ? 26 : 0); }
};
// S4 out of one buffer [t0][t1][p0][p1]: k-tile 0 of all four tiles first (h2[1]'s split hides there), then k-tile 1 in the
// order p0, t0, p1, t1
struct QhS4 : QhNoStores {
    static constexpr int N = 16;
    static constexpr int btile(int acc) { return 8 + (acc == 0 ? 2 : (acc == 1 ? 0 : (acc == 2 ? 3 : 1))); }    // P1 + P2 + P3 = 8; accumulators p0, t0, p1, t1
    static constexpr int torder(int j) { return j == 0 ? 2 : (j == 1 ? 0 : (j == 2 ? 3 : 1)); }     // p0, t0, p1, t1 (buffer tiles)
    static constexpr StepDesc at(int i) { const int kt = i >> 3, j = (i >> 1) & 3; return mkstep<2>(j, torder(j), kt, kt, i & 1); }
    //   steps 0-3: relu+split h2[1] | 9-12: sigmoid(p0) | 12-13: coupling of x[2] | 14-15: the first half of its split
    static constexpr int valu(int i) { return i < 4 ? 26 : (i < 9 ? 0 : (i < 12 ? 28 : (i == 12 ? 36 : (i == 13 ? 24 : 22)))); }
};

// the same tables with the stash writes of the STASH instantiation: ReLU bits 8 VALU per quad, a tile's two mask words 10
// and stores(i) = the vector-memory STORE instructions of step i (a row or sigma quad 1, a tile's mask words 2).  They sit late in
// their phase, behind the last piece of the next phase's weight DMA, so that the wait in front of the next barrier can leave them in
// flight: s_waitcnt vmcnt(N), N = the stores of the steps after that piece (vmcnt retires in issue order; a store is acknowledged
// 2-2.5 us after its issue under this kernel's write traffic, and vmcnt(0) there cost 4.8 K cycles per block, tools/stamps_fwd3p.py).
// Every one of these stores is an unconditional straight-line instruction, so N is exact; a smaller N would only wait longer.
struct QsS1a : QhS1a {
    static constexpr int valu(int i) { return (i == 6 || i == 7) ? 20 : (i >= 8 && i < 12 ? 23 : QhS1a::valu(i)); }
    static constexpr int stores(int i) { return (i >= 2 && i < 6) ? 1 : ((i == 6 || i == 7) ? 2 : (i >= 8 && i < 12 ? 1 : 0)); }
};
struct QsS1b : QhS1b {
    static constexpr int valu(int i) { return i < 8 ? (i >= 6 ? 1 : 0) : (i < 14 ? 23 : 22); }
    static constexpr int stores(int i) { return (i >= 6 && i < 14) ? 1 : 0; }
};
struct QsS23 : QhS23 {
    static constexpr int valu(int i) { return (i >= 4 && i < 12) ? (i == 8 ? 44 : 34) : (i >= 13 ? 34 : 0); }
    static constexpr int stores(int i) { return i == 8 ? 2 : 0; }
};
struct QsS4 : QhS4 {
    static constexpr int valu(int i) { return i < 4 ? 34 : ((i >= 6 && i < 9) ? 10 + QhS4::valu(i) : QhS4::valu(i)); }
    static constexpr int stores(int i) { return (i >= 6 && i < 9) ? 2 : ((i >= 9 && i < 13) ? 1 : 0); }
};
// STASH = 2 (also the h dump): one more store per quad of h1 / h2, in the step that splits the quad
struct QxS23 : QsS23 {
    static constexpr int valu(int i) { return QsS23::valu(i) + (((i >= 4 && i < 12) || i >= 13) ? 2 : 0); }
    static constexpr int stores(int i) { return QsS23::stores(i) + (((i >= 4 && i < 12) || i >= 13) ? 1 : 0); }
};
struct QxS4 : QsS4 {
    static constexpr int valu(int i) { return QsS4::valu(i) + (i < 4 ? 2 : 0); }
    static constexpr int stores(int i) { return QsS4::stores(i) + (i < 4 ? 1 : 0); }
};
template <class PH> constexpr int stores_after(int last_dma_step) {
    int n = 0;
    for (int i = last_dma_step + 1; i < PH::N; ++i) n += PH::stores(i);
    return n;
}

template <int WT, int NWAVES, int ST, int STASH = 0>      // STASH: 0 plain, 1 block outputs + activation stash, 2 also the tiled h dump
__global__ __launch_bounds__(64 * NWAVES, 1) void lsnf_fwd3q_kernel(const Fwd3pArgs a) {
    static_assert(!STASH || ST == 2, "the stash tile is a wave's 32 rows");
    using TS1a = std::conditional_t<STASH != 0, QsS1a, QhS1a>;
    using TS1b = std::conditional_t<STASH != 0, QsS1b, QhS1b>;
    using TS23 = std::conditional_t<STASH == 2, QxS23, std::conditional_t<STASH != 0, QsS23, QhS23>>;
    using TS4 = std::conditional_t<STASH == 2, QxS4, std::conditional_t<STASH != 0, QsS4, QhS4>>;
    using C = Fwd3pCfg<WT>;
    static_assert(WT == 2, "lsnf_fwd3q_kernel: f_width <= 64 instantiation");
    constexpr int THREADS = 64 * NWAVES;
    constexpr int HT = 2, NZT = 4, F = C::F;
    constexpr int SLOT = C::SLOT3;                             // 48 KiB
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cst = smem;
    float* const buf0 = smem + a.n_blocks * C::CONST_FLOATS;   // 2 x SLOT
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n = lane & 15, g = lane >> 4;
    // (no half-a-phase stagger of waves 4-7 here: measured 3 % slower in this kernel, tools/stamps_fwd3p.py -- so two weight buffers)
    P_STAMP(0, "s_memtime");
    P_STAMP(50, "s_memrealtime");
    const int n_phases = 4 * a.n_blocks;
    auto phase_src = [&](int k) -> const float* {
        const float* gb = a.panels3 + (size_t)(k >> 2) * C::BLOCK3;
        const int j = k & 3;
        return gb + (j == 0 ? 0 : (j == 1 ? 2 * 4 * F : (j == 2 ? C::OFF3_S2 : C::OFF3_S4)));
    };
    // Barrier k: my pieces of phase k's weights have landed (vmcnt), everybody's have (barrier), and buffer (k + 1) & 1 is free.
    // The next phase's 48 KiB are NOT issued here in one burst (12 / 6 LDS-DMA instructions per wave, each with its m0
    // hand-over: ~1 200 cycles per phase with one wave per SIMD, tools/stamps_fwd3p.py at 16 384 rows) but one piece at the head
    // of a step of phase k (dma_step), where the issue overlaps the step's MFMAs.  The last phase re-fetches its own weights into
    // the free buffer -- no DMA under a branch; the epilogue waits for it before LDS is reused or released.
    // (STASH: `left` = the stash stores of the phase just run that may stay in flight, see stores_after())
    auto sync_issue = [&](int k, auto left) {
#ifdef LSNF_ABL_SKIPSYNC       // timing diagnostic (racy, wrong numbers): phase boundaries (k & 3) in the mask are not synchronised at all
        if ((LSNF_ABL_SKIPSYNC >> (k & 3)) & 1) return;
#endif
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(decltype(left)::value) : "memory");
        __syncthreads();
#ifdef LSNF_FWD3Q_BURST_DMA
        if (k + 1 < n_phases) issue_kib<48, NWAVES>(phase_src(k + 1), buf0 + ((k + 1) & 1) * SLOT, wave, lane);
#endif
    };
    // pieces per wave and phase; one every EVERY steps (STASH: one per step from step 0, so that the steps behind them can hold the stores)
    constexpr int PER_WAVE = 48 / NWAVES, EVERY = STASH ? 1 : 16 / PER_WAVE, LAST_DMA = (PER_WAVE - 1) * EVERY;
    using Left0 = std::integral_constant<int, 0>;
    using LeftS1a = std::integral_constant<int, stores_after<TS1a>(LAST_DMA)>;
    using LeftS1b = std::integral_constant<int, stores_after<TS1b>(LAST_DMA)>;
    using LeftS23 = std::integral_constant<int, stores_after<TS23>(LAST_DMA) + (STASH == 2 ? 1 : 0)>;     // (+ the exposed quad of h2[0] behind S2+S3)
    using LeftS4 = std::integral_constant<int, stores_after<TS4>(LAST_DMA)>;
    const float* dma_src = nullptr; float* dma_dst = nullptr;
    auto dma_arm = [&](int k) {                                          // phase k is about to run: its steps carry phase k+1's pieces
        dma_src = phase_src(k + 1 < n_phases ? k + 1 : k);
        dma_dst = buf0 + ((k + 1) & 1) * SLOT;
    };
    auto dma_step = [&](auto ic) {
#ifndef LSNF_FWD3Q_BURST_DMA
        constexpr int i = decltype(ic)::value;
        if constexpr (i % EVERY == 0 && i / EVERY < PER_WAVE) issue_piece<NWAVES>(dma_src, dma_dst, i / EVERY, wave, lane);
#endif
    };

    const int wbase = (blockIdx.x * NWAVES + wave) * (16 * ST);
    int sample[2] = {0, 0}, rows[2] = {0, 0}; bool live[2] = {false, false};
#pragma unroll
    for (int st = 0; st < ST; ++st) { sample[st] = wbase + 16 * st + n; live[st] = sample[st] < a.B; rows[st] = live[st] ? sample[st] : a.B - 1; }

    // Loop-carried state: xs[0..1] = split of the block input's first half (= the previous block's split v1), xs[2] with the
    // quads 0, 1 of x[2] split, v[2] = x[2] (fp32, until its split is complete), v[3] = v2[1] BEFORE its coupling, p1 / t1 =
    // the previous block's pre-sigmoid / shift for it (block 0: p1 = 80, t1 = 0: sigmoid = 1 exactly, log = 0 -- the identity;
    // a peeled first S1a that splits the input tiles under its MFMAs instead measured no faster: 92.5 vs 91.5 us, and 21 registers more),
    // lsum = the running sum of log2(1 + exp(-p)) over all blocks (scaled once, in the epilogue)
    Tile16 v[NZT];
    SplitTile16 xs[NZT];
    Tile16 p1, t1;
    float lsum[2] = {0.0f, 0.0f};
    {
        // the row loads go out FIRST: the constant blocks below travel through registers (load, wait, LDS store), and with that
        // copy in front the rows were requested one memory round trip (~2 us) later than necessary
        const Tile16 x0 = load_tile16<HT, ST>(0, a.z_in, rows, a.nz, a.half, g, a.vec4);
        const Tile16 x1 = load_tile16<HT, ST>(1, a.z_in, rows, a.nz, a.half, g, a.vec4);
        v[2] = load_tile16<HT, ST>(2, a.z_in, rows, a.nz, a.half, g, a.vec4);
        v[3] = load_tile16<HT, ST>(3, a.z_in, rows, a.nz, a.half, g, a.vec4);
        __builtin_amdgcn_sched_barrier(0);
        issue_kib<48, NWAVES>(phase_src(0), buf0, wave, lane);
        for (int i = tid; i < a.n_blocks * C::CONST_FLOATS; i += THREADS) cst[i] = a.consts[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) { split_q16<false, ST>(x0, q, xs[0]); split_q16<false, ST>(x1, q, xs[1]); }     // (not hidden: once per launch)
        split_q16<false, ST>(v[2], 0, xs[2]); split_q16<false, ST>(v[2], 1, xs[2]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p1.q[q][r] = 80.0f; t1.q[q][r] = 0.0f; }
    }
    float ell[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) ell[st] = (st < ST && a.objective) ? a.objective[rows[st]] : 0.0f;
    float ss01[2] = {0.0f, 0.0f};
    // STASH: per-lane byte offsets of the stash writes (rows of this wave in a (B, nz) block; this wave's stash tile), and the byte
    // counts of the buffer descriptors
    StashLane sl0 = {0u, 0u, 0u, 0u};
    const LsnfActLayout al = lsnf_act_layout(a.B, HT, WT);
    const unsigned zbytes = (unsigned)a.B * (unsigned)a.nz * 4u, abytes = (unsigned)(al.per_block * 4);
    if constexpr (STASH) {
        const unsigned tile_byte = (unsigned)(wbase >> 5) * (unsigned)(al.per_tile * 4), L = (unsigned)(n + 32 * (g & 1));
        sl0.zoff = ((unsigned)(wbase + n) * (unsigned)a.nz + 4u * g) * 4u;
        sl0.soff = tile_byte + ((unsigned)(g >> 1) * 64u + L) * 16u;
        sl0.moff = tile_byte + (unsigned)(al.mask_off * 4) + L * 4u;
        sl0.msh = 4u * (g >> 1);
        sl0.hoff = (unsigned)(wbase >> 5) * (unsigned)(32 * a.width * 4) + (unsigned)(n * 4 + g) * 16u;
    }
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const unsigned hbytes = (unsigned)(((size_t)a.B + 31) / 32 * 32 * a.width * 4);
    sync_issue(0, Left0{});
    P_STAMP(1, "s_memtime");

    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const float* cb = cst + blk * C::CONST_FLOATS;
        const bool more = blk + 1 < a.n_blocks;
        const int k0 = 4 * blk;
        SplitTile16 vh[4];                          // S2+S3 inputs: split v[0], v[1], h1[0], h1[1]
        SplitTile16 h2s[WT];
        Tile16 hh[4], tp[4];                        // hh: h1[0], h1[1], h2[0], h2[1];  tp: p0, t0, p1, t1
        // STASH: where this block's output rows / stash tiles go, and the previous block's (whose second half is finished under this
        // block's S1a).  Block 0 has no predecessor: those stores go to rows of z_out and to block 0's own tiles, both rewritten later
        // by the same wave; the last block of the call writes its rows to z_out.
        // A part of the stash the caller did not ask for (z_saved or act_saved NULL) gets a descriptor of zero bytes: same instructions,
        // every store dropped.
        const bool zs = STASH && a.z_saved != nullptr, as = STASH && a.act_saved != nullptr;
        const float* zc = more ? (zs ? a.z_saved + (size_t)blk * a.B * a.nz : a.z_out) : a.z_out;
        const float* zp = blk > 0 ? (zs ? a.z_saved + (size_t)(blk - 1) * a.B * a.nz : a.z_out) : a.z_out;
        const StashRsrc rs_zc = stash_rsrc(zc, (more && !zs) ? 0u : zbytes, (unsigned)a.half * 4u);
        const StashRsrc rs_zp = stash_rsrc(zp, (blk > 0 && !zs) ? 0u : zbytes, (unsigned)a.half * 4u);
        const StashRsrc rs_ac = stash_rsrc(as ? a.act_saved + (size_t)blk * al.per_block : a.z_out, as ? abytes : 0u, 4096u);
        const StashRsrc rs_ap = stash_rsrc(as ? a.act_saved + (size_t)(blk > 0 ? blk - 1 : 0) * al.per_block : a.z_out, as ? abytes : 0u, 4096u);
        // (opaque per block: the constant parts of the store offsets then fold into the instructions' immediate fields instead of being
        //  formed once, ahead of the loop, in thirty registers)
        StashLane sl = sl0;
        if constexpr (STASH) asm volatile("" : "+v"(sl.zoff), "+v"(sl.soff), "+v"(sl.moff));
        if constexpr (STASH == 2) asm volatile("" : "+v"(sl.hoff));
        const float* hd = (STASH == 2) ? a.hdump + (size_t)blk * dl.per_block : a.z_out;
        const StashRsrc rs_h1 = stash_rsrc(STASH == 2 ? hd + dl.off_h1 : hd, STASH == 2 ? hbytes : 0u, 4096u);
        const StashRsrc rs_h2 = stash_rsrc(STASH == 2 ? hd + dl.off_h2 : hd, STASH == 2 ? hbytes : 0u, 4096u);
        unsigned mk1[2][2], mk2[2][2], mk3[2][2];   // ReLU bits of h1[1], h2[0], h2[1] until their words are stored (early in S4: see sync_issue)

        if (blk == 1) P_STAMP(10, "s_memtime");
        // ---- S1a: v[0,1]  (model.py:187; actnorm :244,268 folded); carries the end of the previous block's coupling (:414-418) ----
        {
            const float* bv = bias_lane_ptr(cb, g);
            if (k0 > 0) sync_issue(k0, LeftS4{});
            dma_arm(k0);
            run_phase16<TS1a, ST>(v, bv, xs, buf0 + (k0 & 1) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (STASH && i >= 8 && i < 12) stash_row_quad<HT>(rs_zp, sl, 2, i - 8, v[2], a.nz, a.half, g);   // x[2] of the previous block
                if constexpr (i < 2) split_q16<false, ST>(v[2], 2 + i, xs[2]);                 // x[2]: k-tile 2 starts at step 8
                else if constexpr (i < 6) {
                    sigmoid_q16<ST>(p1, i - 2, lsum);
                    if constexpr (STASH) stash_sigma_quad(rs_ap, sl, 1, i - 2, p1);
                } else if constexpr (i < 8) {
                    couple_q16<ST>(v[3], t1, p1, 2 * (i - 6)); couple_q16<ST>(v[3], t1, p1, 2 * (i - 6) + 1);
                    if constexpr (STASH) { stash_row_quad<HT>(rs_zp, sl, 3, 2 * (i - 6), v[3], a.nz, a.half, g); stash_row_quad<HT>(rs_zp, sl, 3, 2 * (i - 6) + 1, v[3], a.nz, a.half, g); }
                } else if constexpr (i < 12) split_q16<false, ST>(v[3], i - 8, xs[3]);         // x[3]: k-tile 3 starts at step 12
            }, dma_step);
        }
        if (blk == 1) P_STAMP(11, "s_memtime");
        // ---- S1b: v[2,3]; split v[0], v[1]: S2's input AND the next block's x[0], x[1] ----
        {
            const float* bv = bias_lane_ptr(cb, g);
            sync_issue(k0 + 1, LeftS1a{});
            dma_arm(k0 + 1);
            run_phase16<TS1b, ST>(v + 2, bv, xs, buf0 + ((k0 + 1) & 1) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (STASH && i >= 6 && i < 14) stash_row_quad<HT>(rs_zc, sl, (i - 6) >> 2, (i - 6) & 3, v[(i - 6) >> 2], a.nz, a.half, g);   // the v1 half of this block's output
                if constexpr (i >= 8) split_q16<false, ST>(v[(i - 8) >> 2], i & 3, vh[(i - 8) >> 2]);    // (xs[0], xs[1] are dead by now: k order)
            }, dma_step);
        }
        if (blk == 1) P_STAMP(12, "s_memtime");
        if (!more) {             // last block: the v1 half is final (model.py:422) -- its stores drain under S2..S4
            // (row indices and addresses are formed HERE from the lane id, not in the prologue and carried through the stack)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int smp[2] = {wbase + (ln & 15), wbase + 16 + (ln & 15)};
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                if constexpr (!STASH) store_tile16<HT, ST>(t, v[t], a.z_out, smp, live, a.nz, a.half, g, a.vec4);    // (STASH: stored under S1b)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int st = 0; st < ST; ++st)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ss01[st] += v[t].q[2 * ft + st][r] * v[t].q[2 * ft + st][r];
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) { ell[st] = ell[st] + cb[32 * C::NP + 0]; ell[st] = ell[st] + cb[32 * C::NP + 1]; }
        // ---- S2 + S3: h1 = relu(actnorm(v1 @ W1)), h2 = relu(actnorm(h1 @ W2))  (model.py:326-328,307-308) ----
        {
            static_assert(C::P1 == 4 && C::P2 == 2 && C::P3 == 2, "bias tile indices of the phase tables");
            const float* bv = bias_lane_ptr(cb, g);
            sync_issue(k0 + 2, LeftS1b{});
            dma_arm(k0 + 2);
            run_phase16<TS23, ST>(hh, bv, vh, buf0 + ((k0 + 2) & 1) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i >= 4 && i < 8) split_q16<true, ST>(hh[0], i - 4, vh[2]);      // h1[0] under h1[1]'s steps
                if constexpr (i >= 8 && i < 12) split_q16<true, ST>(hh[1], i - 8, vh[3]);     // h1[1] under S3's k-tile 0
                if constexpr (i >= 13) split_q16<true, ST>(hh[2], i - 13, h2s[0]);             // h2[0]: its halves complete after steps 12, 13
                if constexpr (STASH) {
                    if constexpr (i >= 4 && i < 8) mask_q16(hh[0], i - 4, mk3);               // (mk3 is free until S4)
                    if constexpr (i == 8) stash_mask_words(rs_ac, sl, 0, mk3);
                    if constexpr (i >= 8 && i < 12) mask_q16(hh[1], i - 8, mk1);
                    if constexpr (i >= 13) mask_q16(hh[2], i - 13, mk2);
                }
                if constexpr (STASH == 2) {
                    if constexpr (i >= 4 && i < 8) stash_h_quad(rs_h1, sl, 0, i - 4, hh[0], a.width);
                    if constexpr (i >= 8 && i < 12) stash_h_quad(rs_h1, sl, 1, i - 8, hh[1], a.width);
                    if constexpr (i >= 13) stash_h_quad(rs_h2, sl, 0, i - 13, hh[2], a.width);
                }
            }, dma_step);
        }
        if (blk == 1) P_STAMP(13, "s_memtime");
        xs[0] = vh[0]; xs[1] = vh[1];            // v1 is the next block's first half (model.py:422): its split is kept
        // ---- S4: p0, t0, p1, t1 = fc_zeros(h2), de-interleaved (model.py:347-349,411-413) + coupling (:414-418) ----
        {
            constexpr int B4 = C::P1 + C::P2 + C::P3;
            static_assert(B4 == 8, "bias tile indices of the phase tables");
            const float* bv = bias_lane_ptr(cb, g);
            split_q16<true, ST>(hh[2], 3, h2s[0]);       // (the last quad of h2[0]: exposed, S4's first step needs it)
            if constexpr (STASH) mask_q16(hh[2], 3, mk2);
            if constexpr (STASH == 2) stash_h_quad(rs_h2, sl, 0, 3, hh[2], a.width);
            sync_issue(k0 + 3, LeftS23{});
            dma_arm(k0 + 3);
            run_phase16<TS4, ST>(tp, bv, h2s, buf0 + ((k0 + 3) & 1) * SLOT, lane, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 4) split_q16<true, ST>(hh[3], i, h2s[1]);                    // h2[1] under k-tile 0
                if constexpr (STASH) {
                    if constexpr (i < 4) mask_q16(hh[3], i, mk3);
                    if constexpr (i == 6) stash_mask_words(rs_ac, sl, 1, mk1);
                    if constexpr (i == 7) stash_mask_words(rs_ac, sl, WT + 0, mk2);
                    if constexpr (i == 8) stash_mask_words(rs_ac, sl, WT + 1, mk3);
                }
                if constexpr (STASH == 2 && i < 4) stash_h_quad(rs_h2, sl, 1, i, hh[3], a.width);
                // k-tile 1: p0's halves are complete after steps 8, 9; t0's after 10, 11; p1's after 12, 13; t1's after 14, 15
                if constexpr (i >= 9 && i < 13) { sigmoid_q16<ST>(tp[0], i - 9, lsum); if constexpr (STASH) stash_sigma_quad(rs_ac, sl, 0, i - 9, tp[0]); }
                if constexpr (i == 12) couple_q16<ST>(v[2], tp[1], tp[0], 0);
                if constexpr (i == 13) { couple_q16<ST>(v[2], tp[1], tp[0], 1); couple_q16<ST>(v[2], tp[1], tp[0], 2); couple_q16<ST>(v[2], tp[1], tp[0], 3); }
                if constexpr (i >= 14) split_q16<false, ST>(v[2], i - 14, xs[2]);
            }, dma_step);
            p1 = tp[2]; t1 = tp[3];              // (the rest of the coupling rides under the next block's S1a)
        }
        if (blk == 1) P_STAMP(14, "s_memtime");
    }
    P_STAMP(40, "s_memtime");
    // the last block's x[3] (no S1a follows)
#pragma unroll
    for (int q = 0; q < 4; ++q) { sigmoid_q16<ST>(p1, q, lsum); couple_q16<ST>(v[3], t1, p1, q); }
    if constexpr (STASH) {
        const bool as = a.act_saved != nullptr;
        const StashRsrc rs_al = stash_rsrc(as ? a.act_saved + (size_t)(a.n_blocks - 1) * al.per_block : a.z_out, as ? abytes : 0u, 4096u);
#pragma unroll
        for (int q = 0; q < 4; ++q) stash_sigma_quad(rs_al, sl0, 1, q, p1);
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) ell[st] = ell[st] + -0.6931471805599453f * l16_group_sum(lsum[st]);

    // ---- epilogue: z_out, logdet, ll = -0.5*sum z^2 + log(2pi) + logdet (train.py:317-319) ----
    float ss[2] = {ss01[0], ss01[1]};
    int ln_e = lane;
    asm volatile("" : "+v"(ln_e));
    const int smp_e[2] = {wbase + (ln_e & 15), wbase + 16 + (ln_e & 15)};
#pragma unroll
    for (int t = HT; t < NZT; ++t) {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int st = 0; st < ST; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss[st] += v[t].q[2 * ft + st][r] * v[t].q[2 * ft + st][r];
        store_tile16<HT, ST>(t, v[t], a.z_out, smp_e, live, a.nz, a.half, g, a.vec4);
    }
    float ll[2] = {0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        ll[st] = (-0.5f * l16_group_sum(ss[st]) + 1.8378770664093453f) + ell[st];
        if (live[st] && g == 0) {
            const int smp = smp_e[st];
            a.logdet_out[smp] = ell[st];
            if (a.ll_out) a.ll_out[smp] = ll[st];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the last phase's trailing weight DMA: before LDS is reused below or released)
    if (a.stats) {
        double dl = 0.0, dd = 0.0;
#pragma unroll
        for (int st = 0; st < ST; ++st)
            if (live[st] && g == 0) { dl += (double)ll[st]; dd += (double)ell[st]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); dd += __shfl_xor(dd, o, 64); }
        __syncthreads();
        double* red = reinterpret_cast<double*>(buf0);
        if (lane == 0) { red[2 * wave] = dl; red[2 * wave + 1] = dd; }
        __syncthreads();
        if (wave == 0) {         // (all 64 lanes: lsnf_publish_stats is a wave-level protocol)
            double tl = 0.0, td = 0.0;
            for (int w = 0; w < NWAVES; ++w) { tl += red[2 * w]; td += red[2 * w + 1]; }
            lsnf_publish_stats(a.stats, tl, td, a.B, lane);
        }
    }
    P_STAMP(41, "s_memtime");
    P_STAMP(51, "s_memrealtime");
}

template <int WT, int NWAVES, int ST, int STASH = 0>
hipError_t launch_fwd3q_w(const Fwd3pArgs& a, hipStream_t stream) {
    using C = Fwd3pCfg<WT>;
    const size_t lds = ((size_t)a.n_blocks * C::CONST_FLOATS + 2 * (size_t)C::SLOT3) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_fwd3q_kernel<WT, NWAVES, ST, STASH>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 16 * ST * NWAVES - 1) / (16 * ST * NWAVES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVES), lds, stream, a);
    return hipGetLastError();
}

#ifdef LSNF_EXPERIMENTAL_KERNELS
template <int WT, int NWAVES>
hipError_t launch_fwd3p_w(const Fwd3pArgs& a, hipStream_t stream) {
    using C = Fwd3pCfg<WT>;
    const size_t lds = ((size_t)a.n_blocks * C::CONST_FLOATS + 3 * (size_t)C::SLOT3) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = lsnf_fwd3p_kernel<WT, NWAVES>;
    static unsigned long long lds_ok = 0;
    if (hipError_t e = lsnf_allow_big_lds((const void*)kern, &lds_ok); e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + 32 * NWAVES - 1) / (32 * NWAVES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVES), lds, stream, a);
    return hipGetLastError();
}
#endif
}  // namespace

// host-side dispatcher (called from lsnf_api.hip); hipErrorInvalidValue = this call is not covered (geometry, or the
// backward's stash / block outputs are wanted): the caller falls back to lsnf_fwd3.hip
hipError_t lsnf_launch_forward3p(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                 const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                 float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4, hipStream_t stream) {
#ifndef LSNF_EXPERIMENTAL_KERNELS
    return hipErrorInvalidValue;         // (not in this build)
#else
    if (g.HT != 2 || g.WT != 2 || z_saved != nullptr || act_saved != nullptr) return hipErrorInvalidValue;
    Fwd3pArgs a;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels3 = plan + g.off_f3_panels + (size_t)first_block * g.f3_block_floats;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4; a.stats = stats;
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    { extern unsigned long long* g_lsnf_stamps;
      if (!g_lsnf_stamps) { if (hipMalloc(&g_lsnf_stamps, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) g_lsnf_stamps = nullptr; }
      a.stamps = g_lsnf_stamps; }
#endif
    return B > 128 * 256 ? launch_fwd3p_w<2, 8>(a, stream) : launch_fwd3p_w<2, 4>(a, stream);
#endif
}

// the 16x16x32 form (lsnf_fwd3q_kernel): the default throughput forward of LSNF_MATH_BF16X3 for calls it covers (nz in 66..128,
// f_width <= 64, no stash / parameter-gradient dump); hipErrorInvalidValue = not covered, the caller falls back to lsnf_fwd3.hip
hipError_t lsnf_launch_forward3q(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                 const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                 float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4, hipStream_t stream,
                                 float* hdump, int hdump_tiled) {
    if (g.HT != 2 || g.WT != 2) return hipErrorInvalidValue;
    // the h dump of the parameter gradients: in its tiled form only, with both parts of the stash, for the whole stack
    if (hdump && (!hdump_tiled || !act_saved || (n_blocks > 1 && !z_saved) || first_block != 0 || (g.width & 15) ||
                  ((size_t)B + 32) * g.width * 4 >= (1ull << 31))) return hipErrorInvalidValue;
    // with the stash, or a part of it (STASH instantiation: buffer stores of whole 16-byte groups through 32-bit offsets): rows that take
    // 16-byte accesses, and a block of rows / of stash tiles below 2 GiB
    const bool stash = act_saved != nullptr || z_saved != nullptr;
    if (stash && (vec4 != 4 || (size_t)B * g.nz * 4 >= (1ull << 31) || lsnf_act_layout(B, g.HT, g.WT).per_block * 4 >= (1ull << 31) ||
                  (((size_t)act_saved | (size_t)z_saved | (size_t)z_out) & 15)))
        return hipErrorInvalidValue;
    Fwd3pArgs a;
    a.consts = plan + g.off_fwd_const + (size_t)first_block * g.fwd_const_floats;
    a.panels3 = plan + g.off_f3b_panels + (size_t)first_block * g.f3_block_floats;
    a.z_in = z_in; a.objective = objective; a.z_out = z_out; a.logdet_out = logdet_out; a.ll_out = ll_out;
    a.B = B; a.nz = g.nz; a.half = g.half; a.n_blocks = n_blocks; a.vec4 = vec4; a.stats = stats;
    a.z_saved = z_saved; a.act_saved = act_saved ? act_saved + (size_t)first_block * lsnf_act_layout(B, g.HT, g.WT).per_block : nullptr;
    a.hdump = hdump; a.width = g.width;
    a.stamps = nullptr;
#ifdef LSNF_STAMPS
    { extern unsigned long long* g_lsnf_stamps;
      if (!g_lsnf_stamps) { if (hipMalloc(&g_lsnf_stamps, sizeof(unsigned long long) * 64 * 4 * 4096) != hipSuccess) g_lsnf_stamps = nullptr; }
      a.stamps = g_lsnf_stamps; }
#endif
    // workgroup shape by batch size (one workgroup per CU; 256 CUs): 256 rows (8 waves x 32) above 32 768 rows; below, 16 rows
    // per wave so that the grid still covers the chip -- 8 waves x 16 rows down to 16 384 rows, 4 waves x 16 rows below
    static const char* shape = getenv("LSNF_FWD3Q_SHAPE");     // experiment knob (tools/shard_times.py): "82", "42", "81", "41"
    const int sh = shape ? atoi(shape) : (B > 128 * 256 ? 82 : (B > 64 * 256 ? 81 : 41));
    if (hdump) return (sh == 82) ? launch_fwd3q_w<2, 8, 2, 2>(a, stream) : launch_fwd3q_w<2, 4, 2, 2>(a, stream);
    if (stash) return (sh == 82) ? launch_fwd3q_w<2, 8, 2, 1>(a, stream) : launch_fwd3q_w<2, 4, 2, 1>(a, stream);
    if (sh == 82) return launch_fwd3q_w<2, 8, 2>(a, stream);
    if (sh == 42) return launch_fwd3q_w<2, 4, 2>(a, stream);
    if (sh == 81) return launch_fwd3q_w<2, 8, 1>(a, stream);
    return launch_fwd3q_w<2, 4, 1>(a, stream);
}
