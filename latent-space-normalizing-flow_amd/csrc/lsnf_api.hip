// lsnf_api.hip -- the C ABI of liblsnf_flow.so (declared in include/lsnf_flow.h).
// Argument validation happens here, on the host, BEFORE any kernel is launched: operand shapes
// and alignments are checked against what the kernels and their grids assume.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <initializer_list>
#include <stdlib.h>
#ifndef LSNF_SMALL_MAX_DEFAULT
#define LSNF_SMALL_MAX_DEFAULT 16384
#endif

#include "../../include/lsnf_flow.h"
#include "lsnf_layout.h"

// Research builds (-DLSNF_EXPERIMENTAL_KERNELS) also carry the bf16x3 scheme on v_mfma_f32_32x32x16_bf16 -- phase-separated (2)
// and software-pipelined (4) -- both measured slower than the 16x16x32 kernels (profiles/HISTORY.md); not part of the ABI.
#define LSNF_MATH_X_BF16X3_32 2
#define LSNF_MATH_X_BF16X3_PIPE 4
#ifdef LSNF_EXPERIMENTAL_KERNELS
#define LSNF_HAVE_X 1
#else
#define LSNF_HAVE_X 0
#endif

// kernel launchers (other translation units)
hipError_t lsnf_launch_prepare(const LsnfGeo& g, const float* const* params_host, float* plan, void* scratch, hipStream_t stream);
size_t lsnf_prep_scratch_bytes(int nz, int depth);
hipError_t lsnf_launch_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                               const float* z_in, const float* objective, float* z_out, float* logdet_out,
                               float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                               hipStream_t stream);
hipError_t lsnf_launch_forward3(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                int shape16, int fixup, hipStream_t stream, float* hdump = nullptr, int hdump_tiled = 0);
hipError_t lsnf_launch_forward3q(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                 const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                 float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4, hipStream_t stream,
                                 float* hdump = nullptr, int hdump_tiled = 0);
hipError_t lsnf_launch_forward3p(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                 const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                 float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4, hipStream_t stream);
hipError_t lsnf_launch_forward2h(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                 const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                 float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                 int shape16, int fixup, hipStream_t stream, float* hdump = nullptr, int hdump_tiled = 0);
hipError_t lsnf_launch_small3_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                      const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                      float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                      hipStream_t stream, float* hdump = nullptr);
hipError_t lsnf_launch_small3_restash(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                      float* act_saved, int vec4, hipStream_t stream);
hipError_t lsnf_launch_small_forward(const LsnfGeo& g, const float* plan, int first_block, int n_blocks, int B,
                                     const float* z_in, const float* objective, float* z_out, float* logdet_out,
                                     float* ll_out, float* z_saved, float* act_saved, double* stats, int vec4,
                                     hipStream_t stream);
hipError_t lsnf_launch_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                               float* z_out, float* objective_out, int vec4, hipStream_t stream);
hipError_t lsnf_launch_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                  const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                  float* g_z_in, float* dump, float* gl_total, int vec4, hipStream_t stream,
                                  const LsnfLangevinArgs* lv = nullptr, const float* act_saved = nullptr);
hipError_t lsnf_launch_small3_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                      float* z_out, float* objective_out, int vec4, hipStream_t stream);
hipError_t lsnf_launch_reverse3(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                float* z_out, float* objective_out, int vec4, int fixup, hipStream_t stream);
hipError_t lsnf_launch_reverse2h(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                 float* z_out, float* objective_out, int vec4, int fixup, hipStream_t stream);
hipError_t lsnf_launch_small_reverse(const LsnfGeo& g, const float* plan, int B, const float* z_in, const float* objective,
                                     float* z_out, float* objective_out, int vec4, hipStream_t stream);
hipError_t lsnf_launch_small_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                        const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale, float* g_z_in,
                                        int vec4, hipStream_t stream, const LsnfLangevinArgs* lv, const float* act_saved,
                                        float* dump = nullptr, float* gl_total = nullptr);
hipError_t lsnf_launch_small3_backward_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                         const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                                         float ll_scale, float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                         float* dump = nullptr, float* gl_total = nullptr);
hipError_t lsnf_launch_backward3_z(const LsnfGeo& g, const float* plan, int B, const float* z_out, const float* z_saved,
                                   const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                                   float* g_z_in, int vec4, hipStream_t stream, const LsnfLangevinArgs* lv,
                                   float* dump = nullptr, float* gl_total = nullptr, int dump_tiled = 0);
bool lsnf_contract_x3_covers(int B, int nz, int half, int width, const float* z_in, const float* z_out, const float* z_saved);
hipError_t lsnf_launch_backward_params(const LsnfGeo& g, const float* plan, const float* const* params_host,
                                       float* const* grads_host, int B, const float* z_in, const float* z_out,
                                       const float* z_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                                       float ll_scale, float* g_z_in, float* workspace, int vec4, int small_batch,
                                       hipStream_t stream, const float* act_saved);

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(LSNF_E_HIP, "%s: %s", what, hipGetErrorString(e));
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
// vector width of the latent-row accesses: every (B, nz) tensor of the call must allow it (NULL pointers do)
int row_vector_width(const LsnfGeo& g, std::initializer_list<const void*> rows) {
    bool a16 = true, a8 = true;
    for (const void* p : rows) { a16 = a16 && aligned16(p); a8 = a8 && aligned8(p); }
    if (g.half % 4 == 0 && a16) return 4;
    if (g.half % 2 == 0 && a8) return 2;
    return 1;
}
bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// arithmetic of the GEMMs (LSNF_MATH=fp32|bf16x3|bf16x3_phased|fp16x2 overrides the default).  The two knobs are
// process-wide settings read by every call: atomics, so that a setter on one thread and a launch on another do not race
// (a launch sees the old or the new value, never a torn one).
std::atomic<int> g_math{-1};
int math_mode() {
    int m = g_math.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("LSNF_MATH");
        m = (e && !strcmp(e, "bf16x3")) ? LSNF_MATH_BF16X3 : (e && !strcmp(e, "bf16x3_phased")) ? LSNF_MATH_BF16X3_PHASED
          : (e && !strcmp(e, "fp16x2")) ? LSNF_MATH_FP16X2 : (e && !strcmp(e, "fp32")) ? LSNF_MATH_FP32
          : (LSNF_HAVE_X && e && !strcmp(e, "bf16x3_32")) ? LSNF_MATH_X_BF16X3_32
          : (LSNF_HAVE_X && e && !strcmp(e, "bf16x3_pipe")) ? LSNF_MATH_X_BF16X3_PIPE : LSNF_MATH_DEFAULT;
        int expected = -1;
        g_math.compare_exchange_strong(expected, m, std::memory_order_relaxed);
        m = g_math.load(std::memory_order_relaxed);
    }
    return m;
}
// rows at or below which the small-batch (latency) kernels are used: LSNF_SMALL_BATCH_AUTO = the measured crossover of
// the arithmetic mode in force -- ONE threshold for the forward, the backward, the Langevin step and the reverse, so the
// kernel family that wrote an activation stash is the family that reads it (LSNF_SMALL_MAX overrides; 0 disables them)
std::atomic<int> g_small_max{-3};                 // -3: not initialised; LSNF_SMALL_BATCH_AUTO; or the rows set by the caller
int small_batch_setting() {
    int v = g_small_max.load(std::memory_order_relaxed);
    if (v == -3) {
        const char* e = getenv("LSNF_SMALL_MAX");
        int init = e ? atoi(e) : LSNF_SMALL_BATCH_AUTO;
        if (init < 0) init = LSNF_SMALL_BATCH_AUTO;
        int expected = -3;
        g_small_max.compare_exchange_strong(expected, init, std::memory_order_relaxed);
        v = g_small_max.load(std::memory_order_relaxed);
    }
    return v;
}
int small_batch_max() {
    const int v = small_batch_setting();
    if (v != LSNF_SMALL_BATCH_AUTO) return v;
    // fp16x2: its throughput forward is the faster one from ~12 K rows (profiles/r01_i_crossover.txt: 40.8 vs 41.9 us at
    // 12 288, 42.5 vs 51.6 at 16 384); every other mode: ~18-20 K (profiles/r01_g_crossover.txt)
    return math_mode() == LSNF_MATH_FP16X2 ? 12288 : LSNF_SMALL_MAX_DEFAULT;
}
// modes whose latency / backward / reverse kernels are the bf16x3 "L16" ones
bool l16_math() { const int m = math_mode(); return m == LSNF_MATH_BF16X3 || m == LSNF_MATH_FP16X2 || m == LSNF_MATH_X_BF16X3_PIPE || m == LSNF_MATH_BF16X3_PHASED; }

int geo_or_fail(LsnfGeo* g, int nz, int width, int depth, int coupling) {
    if (lsnf_geo_init(g, nz, width, depth, coupling))
        return fail(LSNF_E_GEOMETRY, "unsupported geometry nz=%d width=%d depth=%d coupling=%d "
                    "(need nz even in [2,128], width in [1,128], depth in [1,%d], coupling 0 or 1)",
                    nz, width, depth, coupling, LSNF_MAX_DEPTH);
    return 0;
}
}  // namespace

extern "C" {

int lsnf_abi_version(void) { return LSNF_ABI_VERSION; }

int lsnf_set_small_batch_max(int rows) {
    if (rows == -1) return small_batch_max();                      // query: the threshold in force
    const int prev = small_batch_setting();                        // what was SET: rows, or LSNF_SMALL_BATCH_AUTO
    if (rows >= 0 || rows == LSNF_SMALL_BATCH_AUTO) g_small_max.store(rows, std::memory_order_relaxed);
    return prev;
}
int lsnf_set_math_mode(int mode) {
    const int prev = math_mode();
    if (mode == LSNF_MATH_FP32 || mode == LSNF_MATH_BF16X3 || mode == LSNF_MATH_FP16X2 || mode == LSNF_MATH_BF16X3_PHASED ||
        (LSNF_HAVE_X && (mode == LSNF_MATH_X_BF16X3_32 || mode == LSNF_MATH_X_BF16X3_PIPE)))
        g_math.store(mode, std::memory_order_relaxed);
    return prev;
}
const char* lsnf_last_error(void) { return g_err; }

int lsnf_device_arch(int device, char* buf, size_t buflen) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return fail(LSNF_E_NODEVICE, "no HIP device %d (count %d)", device, n);
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
    if (buf && buflen) { strncpy(buf, prop.gcnArchName, buflen - 1); buf[buflen - 1] = 0; }
    return LSNF_OK;
}

size_t lsnf_plan_floats(int nz, int width, int depth, int coupling) {
    LsnfGeo g;
    if (lsnf_geo_init(&g, nz, width, depth, coupling)) return 0;
    return g.total_floats;
}

size_t lsnf_prepare_scratch_bytes(int nz, int width, int depth) {
    (void)width;
    if (nz < 2 || nz > 128 || depth < 1 || depth > LSNF_MAX_DEPTH) return 0;
    return lsnf_prep_scratch_bytes(nz, depth);
}

int lsnf_prepare(const float* const* params_host, int nz, int width, int depth, int coupling, float* plan,
                 void* scratch, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (!params_host || !plan || !scratch) return fail(LSNF_E_ARG, "lsnf_prepare: NULL argument");
    if (!aligned16(plan) || !aligned16(scratch)) return fail(LSNF_E_ARG, "lsnf_prepare: plan/scratch must be 16-byte aligned");
    for (int i = 0; i < depth * LSNF_PARAMS_PER_BLOCK; ++i)
        if (!params_host[i] || !aligned4(params_host[i]))
            return fail(LSNF_E_ARG, "lsnf_prepare: parameter pointer %d (block %d, slot %d) is NULL or misaligned", i,
                        i / LSNF_PARAMS_PER_BLOCK, i % LSNF_PARAMS_PER_BLOCK);
    hipError_t e = lsnf_launch_prepare(g, params_host, plan, scratch, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "lsnf_prepare launch");
    return LSNF_OK;
}

int lsnf_params_fast_path(void) { return l16_math() ? 1 : 0; }

int lsnf_forward(const float* plan, int nz, int width, int depth, int coupling, int first_block, int n_blocks, int B,
                 const float* z_in, const float* objective, float* z_out, float* logdet_out, float* ll_out,
                 float* z_saved, float* act_saved, float* params_workspace, double* stats, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 0 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_forward: B=%d out of range", B);
    if (B == 0) {   // empty batch: nothing to do (pointers of empty tensors may be NULL)
        if (stats && hipMemsetAsync(stats + 4, 0, 3 * sizeof(double), (hipStream_t)stream) != hipSuccess)
            return fail(LSNF_E_HIP, "lsnf_forward: hipMemsetAsync(stats) failed");
        return LSNF_OK;
    }
    if (!plan || !z_in || !z_out || !logdet_out) return fail(LSNF_E_ARG, "lsnf_forward: NULL argument");
    if (first_block < 0 || n_blocks < 1 || first_block + n_blocks > depth)
        return fail(LSNF_E_ARG, "lsnf_forward: blocks [%d,%d) outside [0,%d)", first_block, first_block + n_blocks, depth);
    if (!aligned16(plan)) return fail(LSNF_E_ARG, "lsnf_forward: plan must be 16-byte aligned");
    if (!aligned4(z_in) || !aligned4(z_out) || !aligned4(logdet_out) || !aligned4(objective) || !aligned4(ll_out) || !aligned4(z_saved))
        return fail(LSNF_E_ARG, "lsnf_forward: tensors must be 4-byte aligned");
    if (B == 0) return LSNF_OK;
    const int vec4 = row_vector_width(g, {z_in, z_out, z_saved});
    if (act_saved && !aligned16(act_saved)) return fail(LSNF_E_ARG, "lsnf_forward: act_saved must be 16-byte aligned");
    if (stats && (reinterpret_cast<uintptr_t>(stats) & 7u)) return fail(LSNF_E_ARG, "lsnf_forward: stats must be 8-byte aligned");
    float* hdump = nullptr;              // h1 / h2 of every block into the parameter-gradient workspace (lsnf_backward_params' fast path)
    if (params_workspace) {
        if (!l16_math()) return fail(LSNF_E_ARG, "lsnf_forward: params_workspace needs a bf16x3-family math mode (lsnf_params_fast_path() == 1)");
        if (first_block != 0 || n_blocks != depth || !act_saved || (depth > 1 && !z_saved))
            return fail(LSNF_E_ARG, "lsnf_forward: params_workspace goes with the whole stack, act_saved and z_saved");
        if (!aligned16(params_workspace)) return fail(LSNF_E_ARG, "lsnf_forward: params_workspace must be 16-byte aligned");
        hdump = params_workspace + 4 + (size_t)depth * lsnf_fold_layout(nz, width).per_block;
    }
    // batch-size dispatch: latency kernel (32 rows per workgroup, stages split over the 4 waves) below the
    // crossover, throughput kernel (128 rows per workgroup, weights shared through LDS) above it
    hipError_t e;
    const int math = math_mode();
    const bool split = math == LSNF_MATH_BF16X3 || math == LSNF_MATH_X_BF16X3_32 || math == LSNF_MATH_X_BF16X3_PIPE || math == LSNF_MATH_BF16X3_PHASED;
    // Calls without a stash that the software-pipelined forward covers cross over at 8 192 rows, not at the common threshold:
    // the latency kernel puts 16 / 32 rows on a workgroup up to 4 096 / 8 192 rows (one round of <= 256 workgroups that each
    // stream the weights once: 14.1 / 20.1 us), lsnf_fwd3q_kernel in its 16-rows-per-wave form takes 30.5 us up to 16 384 rows,
    // and a 64-row latency workgroup 32.6 us (tools/chk_small3_st.py, tools/shard_times.py, profiles/r03_shard_times.txt).
    // Only the AUTO threshold moves (an explicit setting is obeyed); calls that write a stash keep the common threshold, so that
    // the family that wrote it is the family that reads it.
    int small_max = small_batch_max();
    if (z_saved == nullptr && act_saved == nullptr && small_batch_setting() == LSNF_SMALL_BATCH_AUTO && small_max > 8192 &&
        math == LSNF_MATH_BF16X3 && g.HT == 2 && g.WT == 2)
        small_max = 8192;
    // Parameter-gradient dump: from LSNF_X3_MIN_ROWS rows the batch contraction of lsnf_params3.hip may read it and asks the workspace's
    // tag word which form h1 / h2 have -- tiled (whole 1 KiB stores) when the bf16x3 throughput forward writes them, row-major otherwise
    int hdump_tiled = 0;
    if (hdump && B >= LSNF_X3_MIN_ROWS) {
        hdump_tiled = (B > small_max && split && math != LSNF_MATH_X_BF16X3_32 && lsnf_dump_can_tile(nz, width) &&
                       lsnf_contract_x3_covers(B, nz, g.half, width, z_in, z_out, z_saved)) ? 1 : 0;
        if (hipMemsetD32Async((hipDeviceptr_t)(params_workspace + lsnf_params_workspace_tag(nz, width, depth, B)), hdump_tiled, 1,
                              (hipStream_t)stream) != hipSuccess)
            return fail(LSNF_E_HIP, "lsnf_forward: hipMemsetD32Async(workspace tag) failed");
    }
    if (B <= small_max) {
        e = hipErrorInvalidValue;
        if (l16_math())                           // latency forward on the bf16 pipe: 16-sample workgroups (lsnf_small3_fwd.hip)
            e = lsnf_launch_small3_forward(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                           z_saved, act_saved, stats, vec4, (hipStream_t)stream, hdump);
        if (e == hipErrorInvalidValue && !hdump)
            e = lsnf_launch_small_forward(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                          z_saved, act_saved, stats, vec4, (hipStream_t)stream);
    } else {
        e = hipErrorInvalidValue;
        // fp16x2 (opt-in): two fp16 terms per operand, three MFMAs per product (lsnf_fwd2h.hip), followed by the bf16x3 kernel
        // as a fix-up pass that recomputes the workgroups in which a wave met an operand outside fp16's range (the flag
        // travels in logdet_out) and exits at once elsewhere.  Not for in-place calls (the fix-up re-reads the inputs) and
        // not with in-kernel batch sums (a partial recomputation cannot repair them): those run bf16x3 directly.
        const bool fp16_ok = math == LSNF_MATH_FP16X2 && stats == nullptr && z_in != z_out &&
                             (objective == nullptr || objective != logdet_out);
        if (fp16_ok) {
            e = lsnf_launch_forward2h(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                      z_saved, act_saved, nullptr, vec4, 1, 0, (hipStream_t)stream, hdump);
            if (e == hipSuccess)
                e = lsnf_launch_forward3(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                         z_saved, act_saved, nullptr, vec4, 1, /*fixup=*/1, (hipStream_t)stream, hdump);
        }
        if (math == LSNF_MATH_BF16X3 && (!hdump || hdump_tiled))   // vector work software-pipelined under 16x16x32 MFMAs (lsnf_fwd3p.hip, lsnf_fwd3q_kernel)
            e = lsnf_launch_forward3q(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                      z_saved, act_saved, stats, vec4, (hipStream_t)stream, hdump, hdump_tiled);
        if (math == LSNF_MATH_X_BF16X3_PIPE && !hdump) // (research builds) the 32x32x16 kernel with its vector work pipelined under the MFMAs
            e = lsnf_launch_forward3p(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                      z_saved, act_saved, stats, vec4, (hipStream_t)stream);
        if (e == hipErrorInvalidValue && (split || math == LSNF_MATH_FP16X2))   // error-free split on the bf16 matrix pipe (lsnf_fwd3.hip)
            e = lsnf_launch_forward3(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                     z_saved, act_saved, stats, vec4, math != LSNF_MATH_X_BF16X3_32, /*fixup=*/0, (hipStream_t)stream, hdump, hdump_tiled);
        if (e == hipErrorInvalidValue && !hdump)  // fp32 MFMA kernel (also: stacks too deep for fwd3's LDS budget)
            e = lsnf_launch_forward(g, plan, first_block, n_blocks, B, z_in, objective, z_out, logdet_out, ll_out,
                                    z_saved, act_saved, stats, vec4, (hipStream_t)stream);
    }
    if (e != hipSuccess) return hip_fail(e, "lsnf_forward launch");
    return LSNF_OK;
}

size_t lsnf_act_saved_floats(int nz, int width, int depth, int B) {
    LsnfGeo g;
    if (lsnf_geo_init(&g, nz, width, depth, 1) || B < 0) return 0;   // independent of the coupling type
    return (size_t)depth * lsnf_act_layout(B, g.HT, g.WT).per_block;
}

int lsnf_restash(const float* plan, int nz, int width, int depth, int coupling, int B, const float* z_out,
                 const float* z_saved, float* act_saved, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 0 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_restash: B=%d out of range", B);
    if (B == 0) return LSNF_OK;
    if (!plan || !z_out || !act_saved || (depth > 1 && !z_saved)) return fail(LSNF_E_ARG, "lsnf_restash: NULL argument");
    if (!aligned16(plan) || !aligned16(act_saved)) return fail(LSNF_E_ARG, "lsnf_restash: plan / act_saved must be 16-byte aligned");
    if (!aligned4(z_out) || !aligned4(z_saved)) return fail(LSNF_E_ARG, "lsnf_restash: tensors must be 4-byte aligned");
    if (!l16_math()) return fail(LSNF_E_ARG, "lsnf_restash: needs a bf16x3-family math mode (lsnf_params_fast_path() == 1)");
    const hipError_t e = lsnf_launch_small3_restash(g, plan, B, z_out, z_saved, act_saved, row_vector_width(g, {z_out, z_saved}),
                                                    (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "lsnf_restash launch");
    return LSNF_OK;
}

int lsnf_reverse(const float* plan, int nz, int width, int depth, int coupling, int B, const float* z_in,
                 const float* objective, float* z_out, float* objective_out, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 0 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_reverse: B=%d out of range", B);
    if (B == 0) return LSNF_OK;
    if (!plan || !z_in || !z_out) return fail(LSNF_E_ARG, "lsnf_reverse: NULL argument");
    if (!aligned16(plan)) return fail(LSNF_E_ARG, "lsnf_reverse: plan must be 16-byte aligned");
    if (!aligned4(z_in) || !aligned4(z_out) || !aligned4(objective) || !aligned4(objective_out))
        return fail(LSNF_E_ARG, "lsnf_reverse: tensors must be 4-byte aligned");
    if (B == 0) return LSNF_OK;
    const int vec4 = row_vector_width(g, {z_in, z_out});
    hipError_t e = hipErrorInvalidValue;
    if (l16_math()) {        // on the bf16 pipe: lsnf_small3_rev.hip / lsnf_rev3.hip
        // the reverse neither writes nor reads a stash, so under the automatic threshold it takes its own crossover: the 64-row latency
        // form is the faster one up to ~28 K rows (profiles/r03_latency_reverse.txt: 34.2 vs 57.9 us at 16 384, 67.3 vs 60.6 at 32 768)
        int small_max = small_batch_max();
        if (small_batch_setting() == LSNF_SMALL_BATCH_AUTO && math_mode() != LSNF_MATH_FP16X2 && small_max < 24576) small_max = 24576;
        // fp16 two-term split (lsnf_rev2h.hip) + the bf16x3 kernel behind it as the early-exit fix-up pass, as in lsnf_forward
        if (B > small_max && math_mode() == LSNF_MATH_FP16X2 && z_in != z_out && (objective == nullptr || objective != objective_out)) {
            e = lsnf_launch_reverse2h(g, plan, B, z_in, objective, z_out, objective_out, vec4, 0, (hipStream_t)stream);
            if (e == hipSuccess)
                e = lsnf_launch_reverse3(g, plan, B, z_in, objective, z_out, objective_out, vec4, /*fixup=*/1, (hipStream_t)stream);
        }
        if (B > small_max && e == hipErrorInvalidValue)
            e = lsnf_launch_reverse3(g, plan, B, z_in, objective, z_out, objective_out, vec4, /*fixup=*/0, (hipStream_t)stream);
        // (the latency form is also faster than the fp32 throughput reverse where the bf16 throughput form does not fit)
        if (e == hipErrorInvalidValue && small_max > 0)
            e = lsnf_launch_small3_reverse(g, plan, B, z_in, objective, z_out, objective_out, vec4, (hipStream_t)stream);
        if (e == hipErrorInvalidValue && B > small_batch_max() && B <= small_max)       // the latency form does not cover this stack
            e = lsnf_launch_reverse3(g, plan, B, z_in, objective, z_out, objective_out, vec4, /*fixup=*/0, (hipStream_t)stream);
    }
    if (e == hipErrorInvalidValue)      // not taken or not covered: the fp32-MFMA kernels of the batch size's family
        e = (B <= small_batch_max())
            ? lsnf_launch_small_reverse(g, plan, B, z_in, objective, z_out, objective_out, vec4, (hipStream_t)stream)
            : lsnf_launch_reverse(g, plan, B, z_in, objective, z_out, objective_out, vec4, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "lsnf_reverse launch");
    return LSNF_OK;
}

int lsnf_backward_z(const float* plan, int nz, int width, int depth, int coupling, int B, const float* z_out,
                    const float* z_saved, const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode,
                    float ll_scale, float* g_z_in, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 0 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_backward_z: B=%d out of range", B);
    if (B == 0) return LSNF_OK;
    if (!plan || !z_out || !g_z_in || (depth > 1 && !z_saved)) return fail(LSNF_E_ARG, "lsnf_backward_z: NULL argument");
    if (!aligned16(plan)) return fail(LSNF_E_ARG, "lsnf_backward_z: plan must be 16-byte aligned");
    if (!aligned4(z_out) || !aligned4(z_saved) || !aligned4(g_z1) || !aligned4(g_logdet) || !aligned4(g_z_in))
        return fail(LSNF_E_ARG, "lsnf_backward_z: tensors must be 4-byte aligned");
    if (B == 0) return LSNF_OK;
    const int vec4 = row_vector_width(g, {z_out, g_z_in, z_saved, g_z1});
    if (act_saved && !aligned16(act_saved)) return fail(LSNF_E_ARG, "lsnf_backward_z: act_saved must be 16-byte aligned");
    hipError_t e = hipErrorInvalidValue;
    if (B <= small_batch_max() && act_saved && l16_math())     // from the stash, on the bf16 pipe (lsnf_small3_bwd.hip)
        e = lsnf_launch_small3_backward_z(g, plan, B, z_out, z_saved, act_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in,
                                          vec4, (hipStream_t)stream, nullptr);
    else if (act_saved && l16_math() && B > small_batch_max())          // throughput form (lsnf_bwd3.hip)
        e = lsnf_launch_backward3_z(g, plan, B, z_out, z_saved, act_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4,
                                    (hipStream_t)stream, nullptr);
    if (e == hipErrorInvalidValue)      // not taken or not covered: the fp32-MFMA kernels of the batch size's family
        e = (B <= small_batch_max())
            ? lsnf_launch_small_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, vec4,
                                             (hipStream_t)stream, nullptr, act_saved)
            : lsnf_launch_backward_z(g, plan, B, z_out, z_saved, g_z1, g_logdet, ll_mode, ll_scale, g_z_in, nullptr,
                                       nullptr, vec4, (hipStream_t)stream, nullptr, act_saved);
    if (e != hipSuccess) return hip_fail(e, "lsnf_backward_z launch");
    return LSNF_OK;
}

int lsnf_langevin_step(const float* plan, int nz, int width, int depth, int coupling, int B, const float* z_cur,
                       const float* z_out, const float* z_saved, const float* act_saved, const float* grad_g,
                       const float* noise, const LsnfRng* rng, float step_size, float* z_new, float* gf_norm, float* gg_norm, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 0 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_langevin_step: B=%d out of range", B);
    if (B == 0) return LSNF_OK;
    if (!plan || !z_cur || !z_out || !z_new || (depth > 1 && !z_saved)) return fail(LSNF_E_ARG, "lsnf_langevin_step: NULL argument");
    if (!aligned16(plan)) return fail(LSNF_E_ARG, "lsnf_langevin_step: plan must be 16-byte aligned");
    if (!aligned4(z_cur) || !aligned4(z_out) || !aligned4(z_saved) || !aligned4(grad_g) || !aligned4(noise) || !aligned4(z_new) ||
        !aligned4(gf_norm) || !aligned4(gg_norm))
        return fail(LSNF_E_ARG, "lsnf_langevin_step: tensors must be 4-byte aligned");
    const int vec4 = row_vector_width(g, {z_out, z_saved, z_cur, grad_g, noise, z_new});
    if (act_saved && !aligned16(act_saved)) return fail(LSNF_E_ARG, "lsnf_langevin_step: act_saved must be 16-byte aligned");
    if (noise && rng) return fail(LSNF_E_ARG, "lsnf_langevin_step: pass either a noise tensor or an rng, not both");
    if (rng && rng->offset_dev && (reinterpret_cast<uintptr_t>(rng->offset_dev) & 7u))
        return fail(LSNF_E_ARG, "lsnf_langevin_step: rng->offset_dev must be 8-byte aligned");
    if (rng && rng->row0 < 0) return fail(LSNF_E_ARG, "lsnf_langevin_step: rng->row0 must be >= 0");
    LsnfLangevinArgs lv = {z_cur, grad_g, noise, z_new, gf_norm, gg_norm, step_size,
                           rng ? LsnfRngArgs{rng->seed, rng->offset, rng->offset_dev, rng->row0, 1}
                               : LsnfRngArgs{0ull, 0ull, nullptr, 0ll, 0}};
    hipError_t e = hipErrorInvalidValue;
    if (B <= small_batch_max() && act_saved && l16_math())
        e = lsnf_launch_small3_backward_z(g, plan, B, z_out, z_saved, act_saved, nullptr, nullptr, /*ll_mode=*/1,
                                          /*ll_scale=*/-1.0f, nullptr, vec4, (hipStream_t)stream, &lv);
    else if (act_saved && l16_math() && B > small_batch_max())
        e = lsnf_launch_backward3_z(g, plan, B, z_out, z_saved, act_saved, nullptr, nullptr, /*ll_mode=*/1, /*ll_scale=*/-1.0f,
                                    nullptr, vec4, (hipStream_t)stream, &lv);
    if (e == hipErrorInvalidValue)      // not taken or not covered: the fp32-MFMA kernels of the batch size's family
        e = (B <= small_batch_max())
            ? lsnf_launch_small_backward_z(g, plan, B, z_out, z_saved, nullptr, nullptr, /*ll_mode=*/1, /*ll_scale=*/-1.0f,
                                             nullptr, vec4, (hipStream_t)stream, &lv, act_saved)
            : lsnf_launch_backward_z(g, plan, B, z_out, z_saved, nullptr, nullptr, /*ll_mode=*/1, /*ll_scale=*/-1.0f,
                                       nullptr, nullptr, nullptr, vec4, (hipStream_t)stream, &lv, act_saved);
    if (e != hipSuccess) return hip_fail(e, "lsnf_langevin_step launch");
    return LSNF_OK;
}

size_t lsnf_backward_params_workspace_floats(int nz, int width, int depth, int B) {
    LsnfGeo g;
    if (lsnf_geo_init(&g, nz, width, depth, 1) || B < 0) return 0;   // independent of the coupling type
    return lsnf_params_workspace_floats(nz, width, depth, B);
}

int lsnf_backward_params(const float* plan, const float* const* params_host, float* const* grads_host, int nz, int width,
                         int depth, int coupling, int B, const float* z_in, const float* z_out, const float* z_saved,
                         const float* act_saved, const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                         float* g_z_in, float* workspace, void* stream) {
    LsnfGeo g;
    if (int rc = geo_or_fail(&g, nz, width, depth, coupling)) return rc;
    if (B < 1 || B > (1 << 28)) return fail(LSNF_E_ARG, "lsnf_backward_params: B=%d out of range", B);
    if (!plan || !params_host || !grads_host || !z_in || !z_out || !workspace || (depth > 1 && !z_saved))
        return fail(LSNF_E_ARG, "lsnf_backward_params: NULL argument");
    if (!aligned16(plan) || !aligned16(workspace)) return fail(LSNF_E_ARG, "lsnf_backward_params: plan/workspace must be 16-byte aligned");
    for (int i = 0; i < depth * LSNF_PARAMS_PER_BLOCK; ++i) {
        if (!params_host[i] || !aligned4(params_host[i]) || !aligned4(grads_host[i]))
            return fail(LSNF_E_ARG, "lsnf_backward_params: parameter/gradient pointer %d is NULL or misaligned", i);
    }
    if (!aligned4(z_in) || !aligned4(z_out) || !aligned4(z_saved) || !aligned4(g_z1) || !aligned4(g_logdet) || !aligned4(g_z_in))
        return fail(LSNF_E_ARG, "lsnf_backward_params: tensors must be 4-byte aligned");
    // (z_in too: the batch contraction of block 0 reads its rows with the same vector width)
    const int vec4 = row_vector_width(g, {z_in, z_out, g_z_in, z_saved, g_z1});
    if (act_saved && !aligned16(act_saved)) return fail(LSNF_E_ARG, "lsnf_backward_params: act_saved must be 16-byte aligned");
    hipError_t e = lsnf_launch_backward_params(g, plan, params_host, grads_host, B, z_in, z_out, z_saved, g_z1, g_logdet,
                                               ll_mode, ll_scale, g_z_in, workspace, vec4, B <= small_batch_max(),
                                               (hipStream_t)stream, l16_math() ? act_saved : nullptr);
    if (e != hipSuccess) return hip_fail(e, "lsnf_backward_params launch");
    return LSNF_OK;
}

}  // extern "C"
