/*
 * lsnf_flow.h -- C ABI of the MI355X (gfx950) latent-flow-prior library `liblsnf_flow.so`.
 *
 * The reference (jianwen-xie/Latent-Space-Normalizing-Flow) has no native/FFI layer: its
 * replaceable unit is the Python class `_netF` (reference model.py:460-498).  This header is
 * the boundary a maintainer binds from Python (ctypes) to make `_netF`
 * run on hand-written HIP kernels.  Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / STL types.
 *   - every `const float*` / `float*` below is a DEVICE pointer (HBM) unless the name ends
 *     in `_host`; tensors are dense row-major fp32: z is (B, nz), per-sample vectors are (B,).
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).  Calls are
 *     asynchronous, do not synchronise the device and write only the buffers named as outputs
 *     (the plan is read-only after lsnf_prepare), so they are re-entrant per stream and
 *     hipGraph-capturable (no allocation inside).  The only process-wide state are the two
 *     tuning knobs below (lsnf_set_small_batch_max, lsnf_set_math_mode): atomics, read once per call.
 *   - return value: 0 on success, a negative LSNF_E_* code on failure; `lsnf_last_error()`
 *     returns a thread-local message.  Nothing falls back to a CPU path.
 *
 * Geometry: nz even, 2 <= nz <= 128; 1 <= width <= 128; 1 <= depth <= 16;
 * coupling 1 (affine, reference default train.py:63) or 0 (additive, model.py:407-408; runs on the
 * affine kernels with a neutral scale path, i.e. ~12 % more MFMA work than strictly needed).
 */
#ifndef LSNF_FLOW_H
#define LSNF_FLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSNF_ABI_VERSION 5

#define LSNF_OK 0
#define LSNF_E_ARG (-1)       /* bad argument (NULL pointer, size out of range, misaligned) */
#define LSNF_E_GEOMETRY (-2)  /* nz / width / depth outside what the kernels are built for */
#define LSNF_E_HIP (-3)       /* a HIP runtime call or kernel launch failed */
#define LSNF_E_NODEVICE (-4)  /* no gfx950 device visible */

/* Number of live parameter tensors per coupling block, in this order (names as in the
 * reference's state_dict, prefix `revnet2d_s.0.revnet2d_step_s.{i}.`; shapes row-major):
 *   0 actnorm.b            (nz)         model.py:230
 *   1 actnorm.logs         (nz)         model.py:233
 *   2 invertible_1x1_conv.w(nz, nz)     model.py:177
 *   3 f.fc_1.w             (nz/2, w)    model.py:318
 *   4 f.fc_1.actnorm.b     (w)
 *   5 f.fc_1.actnorm.logs  (w)
 *   6 f.fc_2.w             (w, w)
 *   7 f.fc_2.actnorm.b     (w)
 *   8 f.fc_2.actnorm.logs  (w)
 *   9 f.fc_zeros.w         (w, n_out)   model.py:340   n_out = nz (affine) | nz/2 (additive)
 *  10 f.fc_zeros.b         (n_out)      model.py:341
 *  11 f.fc_zeros.logs      (n_out)      model.py:342
 * (`f.fc_1.b`, `f.fc_2.b` are dead parameters in the reference, model.py:319,327-330, and the
 * `.bias` keys alias `.b`, model.py:231: none of them cross this boundary.) */
#define LSNF_PARAMS_PER_BLOCK 12

int lsnf_abi_version(void);
const char* lsnf_last_error(void);

/* Tuning knob: batches of at most `rows` rows run on the small-batch (latency) kernels, larger ones on the
 * throughput kernels; both compute the same function (results agree to fp32 rounding, not bit for bit).  ONE threshold
 * for every entry point, so the kernel family that writes an activation stash is the one that reads it.
 *   rows >= 0                 : set the threshold (0 disables the latency kernels); returns the previous SETTING
 *   LSNF_SMALL_BATCH_AUTO (-2): back to the built-in crossover of the arithmetic mode in force (16384 rows; 12288 in
 *                               LSNF_MATH_FP16X2) -- the initial state unless the LSNF_SMALL_MAX environment variable is set;
 *                               returns the previous setting
 *   -1                        : query only: returns the threshold in force (a row count)
 * "previous setting" is a row count or LSNF_SMALL_BATCH_AUTO, so `prev = set(x); ...; set(prev)` restores exactly.
 * Under LSNF_SMALL_BATCH_AUTO one kind of call has its own crossover: lsnf_forward WITHOUT z_saved / act_saved in
 * LSNF_MATH_BF16X3 at nz in 66..128, f_width <= 64 switches to the (software-pipelined) throughput kernel above 8192 rows. */
#define LSNF_SMALL_BATCH_AUTO (-2)
int lsnf_set_small_batch_max(int rows);

/* Arithmetic of the GEMMs.  Affects the forward, the backward and the reverse (both kernel families); the parameter-
 * gradient contraction over the batch is fp32 MFMA in every mode:
 *   LSNF_MATH_BF16X3 : (default) both operands split error-free into three bf16 terms (3 x 8 = 24 significand bits,
 *                      fp32's exponent range), six bf16 MFMAs per product with fp32 accumulation, on
 *                      v_mfma_f32_16x16x32_bf16.  Same accuracy class as fp32 MFMA (dropped terms <= 2^-26 |w||x|);
 *                      results agree with LSNF_MATH_FP32 to fp32 rounding, not bit for bit.  Throughput forward calls
 *                      without stash / parameter-gradient dump at nz in 66..128, f_width <= 64 run lsnf_fwd3q_kernel
 *                      (csrc/lsnf_fwd3p.hip: operand split and coupling epilogue software-pipelined between the MFMAs),
 *                      everything else csrc/lsnf_fwd3.hip (separate split / MFMA / epilogue phases).
 *   LSNF_MATH_BF16X3_PHASED : as LSNF_MATH_BF16X3 with every throughput forward on csrc/lsnf_fwd3.hip (comparison, tests).
 *   LSNF_MATH_FP32   : fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere.
 *   LSNF_MATH_FP16X2 : (opt-in; NARROWER than the reference's fp32: 11 + 11 operand bits) throughput forward and reverse
 *                      with both operands split into two fp16 terms, three fp16 MFMAs per product (csrc/lsnf_fwd2h.hip;
 *                      dropped terms <= 2^-22 |w||x|; operands below 2^-3 in magnitude additionally carry an ABSOLUTE
 *                      error of up to 2^-25, because their second term falls into fp16's subnormals -- relative accuracy
 *                      of small-magnitude latents and their gradients is not fp32's); half the matrix work of
 *                      LSNF_MATH_BF16X3.  fp16's range is
 *                      guarded: a wave that meets an operand (or folded weight) at or beyond 65504 flags its first output
 *                      element, and the LSNF_MATH_BF16X3 kernel queued behind the launch recomputes the flagged
 *                      workgroups (an early-exit launch otherwise), so results are finite wherever the fp32
 *                      computation's are.  In-place calls (z_out == z_in) and calls with in-kernel batch sums (`stats`)
 *                      run LSNF_MATH_BF16X3 directly.  Every other kernel (latency kernels, backward) as LSNF_MATH_BF16X3.
 * mode < 0 only queries.  Returns the previous mode (default LSNF_MATH_DEFAULT, or the LSNF_MATH environment
 * variable "fp32" / "bf16x3" / "bf16x3_phased" / "fp16x2").  (Values 2 and 4 -- the same scheme on v_mfma_f32_32x32x16_bf16, phase-separated /
 * software-pipelined, both measured slower -- exist only in research builds, -DLSNF_EXPERIMENTAL_KERNELS; refused otherwise.) */
#define LSNF_MATH_FP32 0
#define LSNF_MATH_BF16X3 1
#define LSNF_MATH_FP16X2 3
#define LSNF_MATH_BF16X3_PHASED 5
#define LSNF_MATH_DEFAULT LSNF_MATH_BF16X3
int lsnf_set_math_mode(int mode);

/* Device query: writes the gfx arch name (e.g. "gfx950") of device `device`; LSNF_E_NODEVICE
 * if there is none.  Only call that touches the device without doing work. */
int lsnf_device_arch(int device, char* buf, size_t buflen);

/* ---- prepared weights ("plan") ---------------------------------------------------------
 * Batch-independent work of one `_netF` evaluation, hoisted out of the per-call path:
 * exp(3*logs) folding of the three actnorms (model.py:264-268), de-interleave of fc_zeros'
 * shift / scale columns (model.py:411-413), log|det W| in float64 (model.py:182), W^-1
 * (model.py:193), all re-laid-out in MFMA-fragment order.  Valid until a parameter changes.
 * Written by lsnf_prepare only: every other entry point reads it (`const float* plan` means const), so one plan may be
 * shared by any number of streams and captured graphs.  Do not share one plan buffer between devices. */

/* Size in floats of the prepared-weight buffer for this geometry (0 on bad geometry). */
size_t lsnf_plan_floats(int nz, int width, int depth, int coupling);

/* Size in bytes of the scratch `lsnf_prepare` needs (float64 LU workspace). */
size_t lsnf_prepare_scratch_bytes(int nz, int width, int depth);

/* params_host: HOST array of depth*LSNF_PARAMS_PER_BLOCK DEVICE pointers (order above).
 * plan: device buffer of lsnf_plan_floats() floats, 16-byte aligned.  scratch: device. */
int lsnf_prepare(const float* const* params_host, int nz, int width, int depth, int coupling,
                 float* plan, void* scratch, void* stream);

/* ---- forward: replaces `_netF.forward(z, objective)` (model.py:473-483) -----------------
 * Runs blocks [first_block, first_block+n_blocks) of the stack (model.py:357-360; one block =
 * revnet2d_step.forward model.py:391-422) on B rows in ONE launch.
 *   z_in      (B, nz)   input latents
 *   objective (B) or NULL (= zeros)      running log-det in
 *   z_out     (B, nz)   output of the last block run
 *   logdet_out(B)       objective + sum of the blocks' log|det J|
 *   ll_out    (B) or NULL: -0.5*sum_j z_out^2 + log(2*pi) + logdet_out  (train.py:317-319)
 *   z_saved   NULL, or ((n_blocks-1), B, nz): outputs of all but the last block, kept for
 *             lsnf_backward_z / lsnf_backward_params.
 *   act_saved NULL, or lsnf_act_saved_floats(nz,width,depth,B) floats (16-byte aligned): opaque stash of
 *             what autograd would keep besides the block outputs (the coupling's sigmoid and the two
 *             relu masks, model.py:307-308,415), indexed by absolute block.  Handing it to
 *             lsnf_backward_z / lsnf_langevin_step removes their recomputation of the coupling MLP.
 *   params_workspace NULL, or the workspace of the lsnf_backward_params call that will follow for this evaluation
 *             (lsnf_backward_params_workspace_floats(), 16-byte aligned): the forward then also writes the hidden
 *             activations h1, h2 of every block into it, which lets lsnf_backward_params run FROM THE STASH instead of
 *             recomputing the coupling MLP.  Whole stack only, with act_saved and z_saved; needs a bf16x3-family math
 *             mode (lsnf_params_fast_path() == 1, LSNF_E_ARG otherwise).
 *   stats     NULL, or LSNF_STATS_DOUBLES doubles (device, 8-byte aligned) that the caller zero-initialises ONCE:
 *             after the launch stats[4] = sum_b ll_b (train.py:320), stats[5] = sum_b logdet_b,
 *             stats[6] = B.  Summed inside the kernel (fp64 atomics, one pair per workgroup, into 64
 *             sub-accumulators stats[8..]; tickets find the last workgroup, which reads them all, publishes and
 *             re-arms them; everything but stats[4..6] is internal and back at zero after the launch).
 *             One stats buffer must not be shared by launches on different streams.            */
#define LSNF_STATS_DOUBLES 264
int lsnf_forward(const float* plan, int nz, int width, int depth, int coupling,
                 int first_block, int n_blocks, int B,
                 const float* z_in, const float* objective,
                 float* z_out, float* logdet_out, float* ll_out, float* z_saved,
                 float* act_saved, float* params_workspace, double* stats, void* stream);

/* floats of lsnf_forward's optional activation stash for a batch of B rows (0 on bad geometry) */
size_t lsnf_act_saved_floats(int nz, int width, int depth, int B);

/* ---- reverse: replaces `_netF.forward(z, objective, reverse=True)` (model.py:484-498,
 * block inverse model.py:424-456).  Functional: inputs are not modified (the reference
 * mutates them in place, model.py:436-438).  objective_out = objective - sum log|det J|
 * (the reference returns its negation when return_obj=True, model.py:498). */
int lsnf_reverse(const float* plan, int nz, int width, int depth, int coupling, int B,
                 const float* z_in, const float* objective,
                 float* z_out, float* objective_out, void* stream);

/* ---- backward w.r.t. z: replaces autograd of train.py:316-323 ---------------------------
 * Given upstream gradients g_z1 = dL/dz_out (B,nz) (NULL = 0), g_logdet = dL/dlogdet_out
 * (B) (NULL = 0) computes g_z_in = dL/dz_in (B,nz).  z_out / z_saved are what lsnf_forward
 * wrote for the same inputs (full stack: first_block = 0, n_blocks = depth); act_saved is
 * NULL (the coupling MLP is recomputed from z_saved) or the stash that forward filled.
 * If ll_mode != 0 the upstream gradient is that of L = ll_scale * sum_b ll_b, i.e.
 * g_z1 = -ll_scale*z_out, g_logdet = ll_scale, and the two pointers are ignored
 * (train.py:320 uses L = -sum ll -> ll_scale = -1). */
int lsnf_backward_z(const float* plan, int nz, int width, int depth, int coupling, int B,
                    const float* z_out, const float* z_saved, const float* act_saved,
                    const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                    float* g_z_in, void* stream);

/* The activation stash of a forward that was run WITHOUT one (act_saved = NULL), rebuilt from its block outputs: the
 * coupling MLP's input is the first half of the block's output (model.py:422), so sigmoid(p) and the two ReLU masks follow
 * from z_out / z_saved alone (S2, S3 and the pre-sigmoid half of S4 per block; blocks in parallel).  lsnf_restash + the
 * from-the-stash lsnf_backward_z replace the recomputing backward on the bf16 matrix pipe; the ABI allocates nothing, so
 * the caller owns the stash (lsnf_act_saved_floats).  Needs a bf16x3-family math mode (LSNF_E_ARG otherwise). */
int lsnf_restash(const float* plan, int nz, int width, int depth, int coupling, int B,
                 const float* z_out, const float* z_saved, float* act_saved, void* stream);

/* ---- fused Langevin update: replaces train.py:317-329 after the forward ---------------------
 * One launch computes g_f = d(-sum ll)/dz (as lsnf_backward_z with ll_mode=1, ll_scale=-1) and applies
 *     z_new = z_cur - 0.5*s^2 * (grad_g + g_f) + s * noise          (train.py:324,326)
 * plus the per-sample gradient norms of train.py:328-329 (the caller takes their mean).
 *   z_cur  (B,nz) current latents (the z that lsnf_forward was run on); z_out / z_saved / act_saved
 *          (act_saved may be NULL) from that forward
 *   grad_g (B,nz) or NULL (= 0): the generator's gradient z_grad_g (train.py:314)
 *   noise  (B,nz) or NULL: N(0,1) draws supplied by the caller (train.py:326 `randn_like`)
 *   rng    NULL, or (only when noise is NULL) the in-kernel generator below; both NULL = no noise, as the
 *          test-time sampler (train.py:624-625)
 *   z_new  (B,nz), may alias z_cur (in-place update);  gf_norm, gg_norm: (B) or NULL. */
typedef struct LsnfRng {
    /* Counter-based N(0,1) noise made inside the update kernel: Philox4x32-10 + Box-Muller, a pure function of
     * (seed, offset, global row, column) -- the same draw whatever the batch size, kernel family or sharding of
     * the rows over GPUs (exact definition: csrc/lsnf_device.h lsnf_noise_tile, restated in
     * oracle/philox_oracle.py).  Use a new `offset` for every Langevin step. */
    unsigned long long seed;
    unsigned long long offset;
    const unsigned long long* offset_dev;   /* NULL, or device counter added to offset (captured graphs) */
    long long row0;                         /* global index of this call's row 0 (sharded chains); usually 0 */
} LsnfRng;

int lsnf_langevin_step(const float* plan, int nz, int width, int depth, int coupling, int B,
                       const float* z_cur, const float* z_out, const float* z_saved, const float* act_saved,
                       const float* grad_g, const float* noise, const LsnfRng* rng, float step_size,
                       float* z_new, float* gf_norm, float* gg_norm, void* stream);

/* ---- backward w.r.t. the parameters: replaces `loss_f.backward()` (train.py:406-411) --------
 * Gradients of L w.r.t. the 12 live tensors of every block (same order as lsnf_prepare), for the
 * upstream gradients described under lsnf_backward_z (train.py:410: L = -mean ll -> ll_mode=1,
 * ll_scale = -1/B).  Includes d log|det W|/dW = W^-T (model.py:182) and the actnorm log-det terms.
 *   params_host : HOST array of depth*12 DEVICE pointers to the raw parameters (read only)
 *   grads_host  : HOST array of depth*12 DEVICE pointers that receive the gradients (a NULL entry
 *                 skips that tensor); shapes as the parameters
 *   z_in        : (B, nz) input of the stack; z_out / z_saved as written by lsnf_forward
 *   act_saved   : NULL -- the backward recomputes the coupling MLP (fp32 MFMA) -- or the stash of the lsnf_forward call of
 *                 this evaluation, which must then ALSO have been given `params_workspace` = this call's `workspace`
 *                 (fast path: backward from the stash on the bf16 matrix pipe, nothing recomputed; same math mode and
 *                 small-batch threshold in force for both calls; ignored when lsnf_params_fast_path() == 0)
 *   g_z_in      : NULL, or (B, nz) to also receive dL/dz_in (same values as lsnf_backward_z)
 *   workspace   : lsnf_backward_params_workspace_floats() floats, 16-byte aligned.  Opaque between the two calls of the fast path
 *                 (the forward records there in which form it left h1 / h2: from 12 288 rows the batch contraction runs on the bf16
 *                 matrix pipe and the large-batch kernels write their arrays tiled); pass the forward's own z_in / z_out / z_saved.
 * Sums over the batch use fp32 atomics when B > 1024 (order, hence last bits, may vary run to run:
 * tests/test_gpu_module.py bounds the spread at B = 65 536 to 2e-6 of each tensor's norm). */
size_t lsnf_backward_params_workspace_floats(int nz, int width, int depth, int B);
/* 1 if the math mode in force offers the from-the-stash fast path of lsnf_backward_params (every bf16x3-family mode) */
int lsnf_params_fast_path(void);
int lsnf_backward_params(const float* plan, const float* const* params_host, float* const* grads_host,
                         int nz, int width, int depth, int coupling, int B,
                         const float* z_in, const float* z_out, const float* z_saved, const float* act_saved,
                         const float* g_z1, const float* g_logdet, int ll_mode, float ll_scale,
                         float* g_z_in, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSNF_FLOW_H */
