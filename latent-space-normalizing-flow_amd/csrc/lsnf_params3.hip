// lsnf_params3.hip -- the batch contraction of the parameter gradients (step 2 of lsnf_params.hip: dM = A^T G summed over the
// samples, plus the column sums of G) on the bf16 matrix pipe, for large batches.  Replaces, together with lsnf_params.hip, the
// autograd of train.py:406-411 into the weights of _netF.
//
// Arithmetic: both operands are split, error-free, into three bf16 terms (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits) and a product
// is the six largest of the nine term products, accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- the scheme of the forward /
// backward kernels (lsnf_l16.h), here with the SAMPLES as the MFMA's k index: fp32-faithful products, fp32 accumulation, as the
// fp32-MFMA contraction of lsnf_params.hip, at 2.6x its matrix rate.  The kernel is then bound by reading the operands once.
//
// Work split: one workgroup = (block, chunk of the batch) and runs the block's four tasks one after the other
//   T0 dWa  = x^T  g_v      (nz x nz)         T1 dW1' = v1^T g_a1   (half x width)
//   T2 dW2' = h1^T g_a2     (width x width)   T3 [dW3s | dW3p] = h2^T [g_t | g_p]   (width x 2 half; h2 is read ONCE)
// so every workgroup has the same work and the grid is one round of ~256 workgroups of 8 waves.  Per stage of 32 samples: four
// producer waves -- 128 threads load the A rows and 128 the G rows (8 rows x 16 bytes per thread, two stages in flight in registers),
// split them and write the three bf16 planes TRANSPOSED into LDS ([feature][32 samples], 80-byte pitch: the 16-byte writes of a stage
// and the 16-byte operand reads of the MFMAs are both bank-conflict-free); four consumer waves split the output tiles by rows.  Results go to the zero-initialised
// folded buffer with float atomics (LsnfFoldLayout), exactly as in lsnf_params.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "lsnf_layout.h"

#ifndef X3_ABL
#define X3_ABL 0          // timing experiments (wrong results): 1 no atomics, 2 no MFMA, 4 no global loads after the first two stages, 8 no split
#endif
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_S = 32;                       // samples per stage = the MFMA's k extent
constexpr int X3_PITCH = 80;                   // bytes per feature row of a plane (64 of data + 16: odd number of 16-byte units)
constexpr int X3_PLANE = 128 * X3_PITCH;       // one bf16 term of one operand: 128 features
constexpr int X3_OPERAND = 3 * X3_PLANE;
constexpr int X3_BUFFER = 2 * X3_OPERAND;      // A and G of one stage
constexpr int X3_LDS = 2 * X3_BUFFER;          // double-buffered: 122 880 bytes

struct X3Args {
    const float* z_in; const float* z_out; const float* z_saved; const float* dump; float* fold;
    int B, nz, half, width, depth, chunk;
    int g_tiled; const int* h_tag;     // the g arrays / (word read on the device) the h arrays in the tiled form of lsnf_l16.h l16_store_tiled
};
struct X3Task {
    const float* A; int lda, K, a_tiled;
    const float* G0; const float* G1; int ldg0, ldg1, N, nsplit, g_tiled;   // G columns [0, nsplit) from G0, [nsplit, N) from G1
    float* C0; float* C1; int ldc0, ldc1; float* cs0; float* cs1;     // columns [0, nsplit) of the result -> C0 (row stride ldc0), the rest -> C1; column sums
};

__device__ __forceinline__ unsigned x3_pk(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}

// One producer wave stages a SEGMENT of an operand: up to 64 columns of one source array, as feature rows feat0.. of the planes.
// Its rows are read with raw BUFFER loads: no branch around a load (with conditional loads the compiler's wait-count bookkeeping
// gave up at the loop head -- s_waitcnt vmcnt(0) in front of every split: no stage in flight), and the descriptor's extent returns
// zeros for rows past the batch; a column group past the segment's width carries the offset 2^31 (always out of range).
// Sample order inside a stage: thread (column group, rg) holds the samples rg + 4 j, j = 0..7, as its run of eight k indices -- for A
// and G alike, so the order is immaterial to the products.  Row-major source: load j covers four consecutive rows x 256 bytes per
// wave; tiled source (l16_store_tiled): 16-byte unit ((fg >> 2) * 8 + j) * 16 + rg * 4 + (fg & 3) of the 32-sample tile, again
// 256-byte runs, and j * 256 bytes between a thread's loads.
struct X3Segment { __amdgpu_buffer_rsrc_t rs; int ld, ncols, feat0, col0, tiled; };
__device__ __forceinline__ X3Segment x3_segment(const float* base, int col0, int ld, int ncols, int feat0, int B, int tiled) {
    X3Segment g;
    const long rows = tiled ? ((long)B + 31) / 32 * 32 : (long)B;
    const long floats = ncols > 0 ? rows * ld - (tiled ? 0 : col0) : 0;
    g.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + ((ncols > 0 && !tiled) ? col0 : 0), 0, (int)(floats * 4), 0x00020000);
    g.ld = ld; g.ncols = ncols; g.feat0 = feat0; g.col0 = col0; g.tiled = tiled;
    return g;
}
// byte offset of this thread's first load of the stage that starts at row m0 (a multiple of 32), and between its loads
__device__ __forceinline__ unsigned x3_voff(const X3Segment& g, int m0, int col, int rg) {
    if (g.tiled) {
        const int fg = (g.col0 + col) >> 2;
        return (unsigned)(m0 * g.ld * 4 + (((fg >> 2) * 8) * 16 + rg * 4 + (fg & 3)) * 16);
    }
    return (unsigned)(((m0 + rg) * g.ld + col) * 4);
}
__device__ __forceinline__ void x3_load(f32x4 (&r)[8], const X3Segment& g, unsigned voff, unsigned jstride) {
#pragma unroll
    for (int j = 0; j < 8; ++j)     // (the whole offset in the VGPR: the range check then does not depend on how a scalar offset would enter it)
        r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.rs, voff + (unsigned)j * jstride, 0, 0));
}
// split the 8 samples x 4 features of this thread and write them as 3 x 4 sixteen-byte runs (8 consecutive samples of one feature)
template <bool SUMS>
__device__ __forceinline__ void x3_split_store(const f32x4 (&r)[8], char* planes /* + feature row and sample run of this thread */, float (&cs)[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = r[j][f];
        if constexpr (SUMS) cs[f] += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        u32x4 p[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = x3_pk(a[2 * i], a[2 * i + 1]);
                p[t][i] = w;
                if (t < 2) { a[2 * i] -= __builtin_bit_cast(float, w << 16); a[2 * i + 1] -= __builtin_bit_cast(float, w & 0xffff0000u); }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4*>(planes + t * X3_PLANE + f * X3_PITCH) = p[t];
    }
}

// one stage of MFMAs of this wave: m-tiles wave + 4*mi, all NTL n-tiles
template <int MPW, int NTL>
__device__ __forceinline__ void x3_mma(f32x4 (&acc)[MPW][NTL], const char* buf, int wave, int lane) {
    const int row = lane & 15, kg = lane >> 4;
    const char* pa = buf + row * X3_PITCH + kg * 16;
    const char* pg = buf + X3_OPERAND + row * X3_PITCH + kg * 16;
    bf16x8 af[MPW][3];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int t = 0; t < 3; ++t) af[mi][t] = *reinterpret_cast<const bf16x8*>(pa + t * X3_PLANE + (wave + 4 * mi) * 16 * X3_PITCH);
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        bf16x8 gf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) gf[t] = *reinterpret_cast<const bf16x8*>(pg + t * X3_PLANE + nt * 16 * X3_PITCH);
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
#define X3_MMA(TA, TG) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][TA], gf[TG], acc[mi][nt], 0, 0, 0);
            X3_MMA(2, 0) X3_MMA(0, 2) X3_MMA(1, 1) X3_MMA(1, 0) X3_MMA(0, 1) X3_MMA(0, 0)      // smallest terms first
#undef X3_MMA
        }
    }
}

// One task.  Waves 4..7 are the PRODUCERS (global loads, split, LDS writes, column sums), waves 0..3 the CONSUMERS (operand reads,
// MFMAs, the atomics at the end): two waves per SIMD, one of each role, so the split's VALU work and the load waits of one run under
// the MFMAs of the other (with all waves doing both in turn, the four parts of a stage simply added up: tools/ablate_x3.sh).
// One barrier per stage: the producers' write of stage s + 2 follows barrier s + 1, which the consumers reach after reading stage s.
template <int MPW, int NTL>
__device__ __forceinline__ void x3_task(const X3Task& t, int m_begin, int m_end, int m_end_all, char* lds) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ns = (m_end - m_begin + X3_S - 1) / X3_S;
    if (wave >= 4) {
        // waves 4, 5: the A columns [0, 64), [64, 128); waves 6, 7: the G columns -- of the two sources, or the two halves of one.
        // Thread = (column group c4 = lane >> 2, row group rg = lane & 3): 8 rows x 4 columns per stage.
        const bool isg = wave >= 6, second = (wave & 1) != 0;
        X3Segment g;
        if (!isg) g = second ? x3_segment(t.A, 64, t.lda, t.K - 64, 64, m_end_all, t.a_tiled) : x3_segment(t.A, 0, t.lda, min(t.K, 64), 0, m_end_all, t.a_tiled);
        else if (t.G1 != nullptr) g = second ? x3_segment(t.G1, 0, t.ldg1, t.N - t.nsplit, t.nsplit, m_end_all, t.g_tiled) : x3_segment(t.G0, 0, t.ldg0, t.nsplit, 0, m_end_all, t.g_tiled);
        else g = second ? x3_segment(t.G0, 64, t.ldg0, t.N - 64, 64, m_end_all, t.g_tiled) : x3_segment(t.G0, 0, t.ldg0, min(t.N, 64), 0, m_end_all, t.g_tiled);
        const int rg = lane & 3, col = 4 * (lane >> 2);
        const bool live = col < g.ncols;                               // (a dead column group writes nothing: those feature rows feed
        char* wr = lds + (isg ? X3_OPERAND : 0) + (g.feat0 + col) * X3_PITCH + rg * 16;   //  output tiles that the epilogue drops)
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 r0[8], r1[8], r2[8];                                     // three stages of rows in flight
        unsigned voff = live ? x3_voff(g, m_begin, col, rg) : 0x80000000u;
        const unsigned step = (unsigned)(X3_S * g.ld * 4), jstride = g.tiled ? 256u : (unsigned)(16 * g.ld);
        x3_load(r0, g, voff, jstride); voff += step;
        x3_load(r1, g, voff, jstride); voff += step;
        x3_load(r2, g, voff, jstride); voff += step;
        auto stage = [&](f32x4 (&r)[8], int s) {                       // stage s: split + write into buffer s & 1, then fetch stage s + 3
            char* w = wr + (s & 1) * X3_BUFFER;
            if (live && (!(X3_ABL & 8) || s == 0)) { if (isg) x3_split_store<true>(r, w, cs); else x3_split_store<false>(r, w, cs); }
            if (!(X3_ABL & 4)) { x3_load(r, g, voff, jstride); voff += step; }
            __syncthreads();
        };
        for (int s = 0; s < ns; s += 3) {                              // (the conditions are workgroup-uniform)
            stage(r0, s);
            if (s + 1 < ns) stage(r1, s + 1);
            if (s + 2 < ns) stage(r2, s + 2);
        }
        __syncthreads();     // (the consumers have read the last stage: the next task may rewrite the buffers)
        if (isg) {           // column sums of G (bias gradients): the four row groups of a column group are adjacent lanes
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float v = cs[f];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                const int n = g.feat0 + col + f;
                if (rg == 0 && live && n < t.N) atomicAdd(n < t.nsplit ? t.cs0 + n : t.cs1 + (n - t.nsplit), v);
            }
        }
        return;
    }
    f32x4 acc[MPW][NTL];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[mi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ns; ++s) {
        __syncthreads();
        if (!(X3_ABL & 2)) x3_mma<MPW, NTL>(acc, lds + (s & 1) * X3_BUFFER, wave, lane);
    }
    __syncthreads();
    // results: acc[mi][nt][r] = dM[16*(wave + 4*mi) + 4*(lane >> 4) + r][16*nt + (lane & 15)]
    const int n0 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int n = 16 * nt + n0;
            if (n < t.N) {
                float* c = n < t.nsplit ? t.C0 + n : t.C1 + (n - t.nsplit);
                const int ldc = n < t.nsplit ? t.ldc0 : t.ldc1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * (wave + 4 * mi) + 4 * g + r;
                    if (k < t.K && (!(X3_ABL & 1) || blockIdx.x == 0)) atomicAdd(c + (size_t)k * ldc, acc[mi][nt][r]);
                }
            }
        }
}

__device__ __forceinline__ void x3_run(const X3Task& t, int m_begin, int m_end, int m_end_all, char* lds) {
    const int mt = (t.K + 15) / 16, nt = (t.N + 15) / 16;          // workgroup-uniform
    if (mt > 4) { if (nt > 4) x3_task<2, 8>(t, m_begin, m_end, m_end_all, lds); else x3_task<2, 4>(t, m_begin, m_end, m_end_all, lds); }
    else { if (nt > 4) x3_task<1, 8>(t, m_begin, m_end, m_end_all, lds); else x3_task<1, 4>(t, m_begin, m_end, m_end_all, lds); }
}

__global__ __launch_bounds__(512, 1) void lsnf_contract_x3_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(16))) char x3_lds[];
    const int blk = blockIdx.y;
    const int m_begin = blockIdx.x * a.chunk, m_end = min(a.B, m_begin + a.chunk);
    const LsnfDumpLayout dl = lsnf_dump_layout(a.B, a.nz, a.width);
    const LsnfFoldLayout fl = lsnf_fold_layout(a.nz, a.width);
    const float* dmp = a.dump + (size_t)blk * dl.per_block;
    float* fold = a.fold + (size_t)blk * fl.per_block;
    const float* yblk = (blk == a.depth - 1) ? a.z_out : a.z_saved + (size_t)blk * a.B * a.nz;
    const float* xblk = (blk == 0) ? a.z_in : a.z_saved + (size_t)(blk - 1) * a.B * a.nz;
    const int h_tiled = *a.h_tag, gt_ = a.g_tiled;                     // (uniform)
    X3Task t;
    // T0: dWa = x^T g_v; tiled dump: g_v = [g_v1 (its own array, half columns) | g_t]
    t.A = xblk; t.lda = a.nz; t.K = a.nz; t.a_tiled = 0; t.g_tiled = gt_; t.N = a.nz;
    if (gt_) { t.G0 = dmp + dl.off_gv; t.G1 = dmp + dl.off_gt; t.ldg0 = a.half; t.ldg1 = a.half; t.nsplit = a.half; }
    else { t.G0 = dmp + dl.off_gv; t.G1 = nullptr; t.ldg0 = a.nz; t.ldg1 = 0; t.nsplit = a.nz; }
    t.C0 = fold + fl.dWa; t.C1 = fold + fl.dWa + t.nsplit; t.ldc0 = a.nz; t.ldc1 = a.nz; t.cs0 = fold + fl.dca; t.cs1 = fold + fl.dca + t.nsplit;
    x3_run(t, m_begin, m_end, a.B, x3_lds);
    // T1: dW1' = v1^T g_a1
    t.A = yblk; t.lda = a.nz; t.K = a.half; t.G0 = dmp + dl.off_ga1; t.G1 = nullptr; t.ldg0 = a.width; t.N = a.width; t.nsplit = a.width;
    t.C0 = fold + fl.dW1; t.ldc0 = a.width; t.cs0 = fold + fl.dc1;
    x3_run(t, m_begin, m_end, a.B, x3_lds);
    // T2: dW2' = h1^T g_a2
    t.A = dmp + dl.off_h1; t.lda = a.width; t.K = a.width; t.a_tiled = h_tiled; t.G0 = dmp + dl.off_ga2; t.ldg0 = a.width; t.N = a.width; t.nsplit = a.width;
    t.C0 = fold + fl.dW2; t.ldc0 = a.width; t.cs0 = fold + fl.dc2;
    x3_run(t, m_begin, m_end, a.B, x3_lds);
    // T3: [dW3s | dW3p] = h2^T [g_t | g_p]
    t.A = dmp + dl.off_h2; t.lda = a.width; t.K = a.width; t.G0 = dmp + dl.off_gt; t.G1 = dmp + dl.off_gp; t.ldg0 = a.half; t.ldg1 = a.half;
    t.N = 2 * a.half; t.nsplit = a.half;
    t.C0 = fold + fl.dW3s; t.C1 = fold + fl.dW3p; t.ldc0 = a.half; t.ldc1 = a.half; t.cs0 = fold + fl.dc3s; t.cs1 = fold + fl.dc3p;
    x3_run(t, m_begin, m_end, a.B, x3_lds);
}

hipError_t x3_allow_lds(const void* kern) {
    static bool done = false;
    if (done) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    if (e == hipSuccess) done = true;
    return e;
}
}  // namespace

// hipErrorInvalidValue = not covered (the caller runs the fp32-MFMA contraction of lsnf_params.hip): rows that do not take 16-byte
// loads, operands wider than 128 features.  (all rows 16-byte aligned: the caller checked the tensors and nz, width, half % 4 == 0)
// Does this kernel take the contraction of such a call?  ONE rule for lsnf_forward (which then may write h1 / h2 tiled) and for
// lsnf_backward_params: batch size, geometry (operands of <= 128 features in whole 16-byte groups, 32-bit byte offsets), the three
// latent tensors 16-byte aligned, and the process-wide knobs LSNF_TN_X3=0 / LSNF_TN_PLAIN (fp32-MFMA kernels of lsnf_params.hip).
bool lsnf_contract_x3_covers(int B, int nz, int half, int width, const float* z_in, const float* z_out, const float* z_saved) {
    static const bool knob = [] { const char* e = getenv("LSNF_TN_X3"); return (e ? atoi(e) != 0 : true) && getenv("LSNF_TN_PLAIN") == nullptr; }();
    if (!knob || B < LSNF_X3_MIN_ROWS) return false;
    if (nz > 128 || width > 128 || 2 * half > 128 || (nz & 3) || (width & 3) || (half & 3) || ((size_t)B + 32) * 128 * 4 >= (1ull << 31)) return false;
    return (((size_t)z_in | (size_t)z_out | (size_t)z_saved) & 15) == 0;
}
hipError_t lsnf_launch_contract_x3(const float* z_in, const float* z_out, const float* z_saved, const float* dump, float* fold,
                                   int B, int nz, int half, int width, int depth, int chunk_override, int g_tiled, const int* h_tag,
                                   hipStream_t stream) {
    if (!lsnf_contract_x3_covers(B, nz, half, width, z_in, z_out, z_saved)) return hipErrorInvalidValue;
    X3Args a;
    a.g_tiled = g_tiled; a.h_tag = h_tag;
    a.z_in = z_in; a.z_out = z_out; a.z_saved = z_saved; a.dump = dump; a.fold = fold;
    a.B = B; a.nz = nz; a.half = half; a.width = width; a.depth = depth;
    // one round of workgroups: ~256 / depth chunks of the batch, in whole stages
    int chunks = 256 / depth;
    if (chunks < 1) chunks = 1;
    int chunk = (B + chunks - 1) / chunks;
    chunk = (chunk + X3_S - 1) / X3_S * X3_S;
    if (chunk_override > 0) chunk = (chunk_override + X3_S - 1) / X3_S * X3_S;
    a.chunk = chunk;
    chunks = (B + chunk - 1) / chunk;
    if (hipError_t e = x3_allow_lds((const void*)lsnf_contract_x3_kernel); e != hipSuccess) return e;
    hipLaunchKernelGGL(lsnf_contract_x3_kernel, dim3(chunks, depth), dim3(512), X3_LDS, stream, a);
    return hipGetLastError();
}
