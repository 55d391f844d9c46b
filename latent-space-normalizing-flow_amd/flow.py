"""Functional host API over the C ABI: prepared weights ("plan") + forward / reverse /
backward of the flow-prior stack on device-resident fp32 tensors.

PyTorch is plumbing here (device memory, streams); every number is produced by the HIP
kernels in csrc/.  All functions raise on CPU tensors: there is no CPU path."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import LSNF_PARAMS_PER_BLOCK, LsnfError

# order of the live tensors of one coupling block at the ABI (include/lsnf_flow.h)
BLOCK_PARAM_KEYS = (
    "actnorm.b", "actnorm.logs", "invertible_1x1_conv.w",
    "f.fc_1.w", "f.fc_1.actnorm.b", "f.fc_1.actnorm.logs",
    "f.fc_2.w", "f.fc_2.actnorm.b", "f.fc_2.actnorm.logs",
    "f.fc_zeros.w", "f.fc_zeros.b", "f.fc_zeros.logs",
)


def block_prefix(i: int, level: int = 0) -> str:
    return f"revnet2d_s.{level}.revnet2d_step_s.{i}."


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _need_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise LsnfError(f"{name} must live on the GPU (got {t.device}); this library has no CPU path")
    if t.dtype != torch.float32:
        raise LsnfError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise LsnfError(f"{name} must be contiguous")


def _check_out(t: torch.Tensor, name: str, need: int, device, dtype=torch.float32, hint: str = ""):
    """A buffer a kernel writes: on the call's device, of the dtype the kernel stores, contiguous, large enough."""
    if not t.is_cuda or t.device != device:
        raise LsnfError(f"{name} must live on {device} (got {t.device})")
    if t.dtype != dtype:
        raise LsnfError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise LsnfError(f"{name} must be contiguous")
    if t.numel() < need:
        raise LsnfError(f"{name} has {t.numel()} elements, the kernel writes {need}" + (f" (allocate it {hint})" if hint else ""))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@dataclass
class FlowPlan:
    """Prepared (folded, padded, fragment-ordered) weights of one `_netF`, device resident."""
    nz: int
    width: int
    depth: int
    coupling: int
    buf: torch.Tensor        # fp32 plan buffer
    scratch: torch.Tensor    # float64 LU workspace (kept so re-preparation allocates nothing)

    @property
    def device(self):
        return self.buf.device

    def logabsdet(self) -> torch.Tensor:
        """(depth,) float64: log|det W_i| computed by the Gauss-Jordan kernel."""
        n = self.nz
        per = n * n + 8
        return self.scratch.view(self.depth, per)[:, n * n].clone()

    def winv(self) -> torch.Tensor:
        """(depth, nz, nz) float64 inverse of every 1x1-conv matrix."""
        n = self.nz
        per = n * n + 8
        return self.scratch.view(self.depth, per)[:, : n * n].reshape(self.depth, n, n).clone()


def alloc_plan(nz: int, width: int, depth: int, coupling: int, device) -> FlowPlan:
    lib = _lib.load()
    nfl = lib.lsnf_plan_floats(nz, width, depth, coupling)
    if nfl == 0:
        raise LsnfError(f"unsupported geometry nz={nz} width={width} depth={depth} coupling={coupling}")
    nsc = lib.lsnf_prepare_scratch_bytes(nz, width, depth)
    buf = torch.empty(nfl, dtype=torch.float32, device=device)
    scratch = torch.empty(nsc // 8, dtype=torch.float64, device=device)
    return FlowPlan(nz, width, depth, coupling, buf, scratch)


def prepare(params: Sequence[torch.Tensor], nz: int, width: int, depth: int, coupling: int = 1,
            plan: Optional[FlowPlan] = None) -> FlowPlan:
    """params: depth*12 device tensors in BLOCK_PARAM_KEYS order per block.
    Replaces the batch-independent work the reference redoes on every call (model.py:182,193,264,349)."""
    lib = _lib.load()
    if len(params) != depth * LSNF_PARAMS_PER_BLOCK:
        raise LsnfError(f"expected {depth * LSNF_PARAMS_PER_BLOCK} parameter tensors, got {len(params)}")
    half = nz // 2
    n_out = nz if coupling == 1 else half
    shapes = [nz, nz, nz * nz, half * width, width, width, width * width, width, width, width * n_out, n_out, n_out]
    for i, t in enumerate(params):
        _need_cuda(t, f"param[{i}]")
        if t.numel() != shapes[i % LSNF_PARAMS_PER_BLOCK]:
            raise LsnfError(f"param[{i}] ({BLOCK_PARAM_KEYS[i % 12]}) has {t.numel()} elements, "
                            f"expected {shapes[i % 12]}")
    dev = params[0].device
    if plan is None:
        plan = alloc_plan(nz, width, depth, coupling, dev)
    arr = (ctypes.c_void_p * len(params))(*[t.data_ptr() for t in params])
    with torch.cuda.device(dev):
        rc = lib.lsnf_prepare(arr, nz, width, depth, coupling, _ptr(plan.buf), _ptr(plan.scratch), _stream_ptr(dev))
    _lib.check(rc, "lsnf_prepare")
    return plan


def params_from_state_dict(sd, depth: int, device=None) -> List[torch.Tensor]:
    """Pick the 12 live tensors per block out of a reference-keyed state_dict."""
    out = []
    for i in range(depth):
        pre = block_prefix(i)
        for k in BLOCK_PARAM_KEYS:
            t = sd[pre + k]
            if device is not None:
                t = t.to(device)
            out.append(t.detach().to(torch.float32).contiguous())
    return out


def forward(plan: FlowPlan, z: torch.Tensor, objective: Optional[torch.Tensor] = None, *,
            first_block: int = 0, n_blocks: Optional[int] = None, want_ll: bool = True,
            save_for_backward: bool = False, out: Optional[Tuple[torch.Tensor, ...]] = None,
            stats: Optional[torch.Tensor] = None, act_saved: Optional[torch.Tensor] = None,
            z_saved_out: Optional[torch.Tensor] = None, params_ws: Optional[torch.Tensor] = None):
    """One launch: blocks [first_block, first_block+n_blocks) on (B, nz) rows.
    Returns (z_out, logdet, ll or None, z_saved or None).  model.py:473-483 + train.py:317-319.
    act_saved: optional buffer from `new_act_saved()`, filled with the sigmoid / relu-mask stash that lets
    `backward_z` / the Langevin step skip recomputing the coupling MLP.
    stats: optional buffer from `new_stats()`; afterwards stats[4] = sum ll, stats[5] = sum logdet, stats[6] = B
    (summed inside the kernel -- no separate reduction launch).
    params_ws: optional workspace from `new_params_workspace()` (with act_saved and the block outputs): the forward also
    writes the hidden activations there, so that `backward_params(..., act_saved=, workspace=)` runs from the stash."""
    lib = _lib.load()
    _need_cuda(z, "z")
    if z.dim() != 2 or z.shape[1] != plan.nz:
        raise LsnfError(f"z must be (B, {plan.nz}), got {tuple(z.shape)}")
    B = z.shape[0]
    n_blocks = plan.depth - first_block if n_blocks is None else n_blocks
    if objective is not None:
        _need_cuda(objective, "objective")
        if objective.numel() != B:
            raise LsnfError("objective must have B elements")
    if out is not None:
        z_out, logdet, ll = out
    else:
        z_out = torch.empty_like(z)
        logdet = torch.empty(B, dtype=torch.float32, device=z.device)
        ll = torch.empty(B, dtype=torch.float32, device=z.device) if want_ll else None
    saved = z_saved_out                     # caller-owned (n_blocks-1, B, nz) buffer for the block outputs, or
    if saved is None and save_for_backward and n_blocks > 1:
        saved = torch.empty((n_blocks - 1, B, plan.nz), dtype=torch.float32, device=z.device)
    # The kernels WRITE these caller-owned buffers without knowing their size (the ABI passes pointers): check them here.
    _check_out(z_out, "out[0] (z_out)", B * plan.nz, z.device)
    _check_out(logdet, "out[1] (logdet)", B, z.device)
    if ll is not None:
        _check_out(ll, "out[2] (ll)", B, z.device)
    if saved is not None:
        _check_out(saved, "z_saved", max(n_blocks - 1, 0) * B * plan.nz, z.device)
    if stats is not None:               # ABI v5: 264 doubles (8 + 64 sub-accumulators of 4); the kernel writes all of them
        _check_out(stats, "stats", STATS_DOUBLES, z.device, torch.float64, "from new_stats()")
    if act_saved is not None:
        _check_out(act_saved, "act_saved", lib.lsnf_act_saved_floats(plan.nz, plan.width, plan.depth, B), z.device,
                   hint="from new_act_saved()")
    if params_ws is not None:
        _check_out(params_ws, "params_ws", lib.lsnf_backward_params_workspace_floats(plan.nz, plan.width, plan.depth, B), z.device,
                   hint="from new_params_workspace()")
    with torch.cuda.device(z.device):
        rc = lib.lsnf_forward(_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, first_block, n_blocks, B,
                              _ptr(z), _ptr(objective), _ptr(z_out), _ptr(logdet), _ptr(ll), _ptr(saved),
                              _ptr(act_saved), _ptr(params_ws), _ptr(stats), _stream_ptr(z.device))
    _lib.check(rc, "lsnf_forward")
    return z_out, logdet, ll, saved


class BoundForward:
    """`forward(plan, z, out=..., stats=...)` with everything but the `stats` row bound and checked ONCE: a call is then one ctypes
    call on the current stream (~4 us of host time instead of ~12: tensor checks, size queries and pointer extraction are what a
    20 us strong-scaling shard launch cannot afford per step, tools/host_cost_allreduce.py).  Same launch, same results.

        fw = BoundForward(plan, z, out)          # buffers as for forward(); the plan is re-read at every call (data_ptr is stable)
        fw(stats_row)                            # == forward(plan, z, out=out, stats=stats_row)
    The caller keeps z / out / the plan alive and on the device that is current at call time."""

    def __init__(self, plan: FlowPlan, z: torch.Tensor, out, objective: Optional[torch.Tensor] = None, act_saved: Optional[torch.Tensor] = None,
                 z_saved_out: Optional[torch.Tensor] = None):
        probe = new_stats(z.device)
        forward(plan, z, objective, out=out, stats=probe, act_saved=act_saved, z_saved_out=z_saved_out)       # every check of forward(), once
        self._keep = (plan, z, out, objective, act_saved, z_saved_out)
        self._lib = _lib.load()
        self._dev = z.device
        B = z.shape[0]
        self._head = (_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, 0, plan.depth, B,
                      _ptr(z), _ptr(objective), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(z_saved_out), _ptr(act_saved), None)

    def __call__(self, stats: Optional[torch.Tensor] = None, stream: Optional[torch.cuda.Stream] = None) -> None:
        """stream: the stream to launch on (default: the current one) -- passing it saves the `with torch.cuda.stream(...)` around the call."""
        if stats is not None and (stats.dtype != torch.float64 or stats.numel() < STATS_DOUBLES or stats.device != self._dev):
            raise LsnfError("stats must be a float64 buffer of LSNF_STATS_DOUBLES doubles on the call's device (new_stats())")
        sp = (torch.cuda.current_stream(self._dev) if stream is None else stream).cuda_stream
        rc = self._lib.lsnf_forward(*self._head, None if stats is None else stats.data_ptr(), sp)
        if rc:
            _lib.check(rc, "lsnf_forward")


SMALL_BATCH_AUTO = -2


def set_small_batch_max(rows: int) -> int:
    """Batches <= rows use the small-batch (latency) kernels (one threshold for every entry point).  rows >= 0 sets it,
    SMALL_BATCH_AUTO goes back to the built-in crossover of the arithmetic mode; both return the previous SETTING (a row
    count or SMALL_BATCH_AUTO), so `prev = set(x); ...; set(prev)` restores exactly.  rows == -1 only queries the threshold
    in force."""
    return _lib.load().lsnf_set_small_batch_max(int(rows))


MATH_FP32, MATH_BF16X3, MATH_FP16X2, MATH_BF16X3_PHASED = 0, 1, 3, 5
# research builds of the library only (make EXTRA=-DLSNF_EXPERIMENTAL_KERNELS; refused by the shipped one): the bf16x3 scheme on
# v_mfma_f32_32x32x16_bf16, phase-separated / software-pipelined -- both measured slower (profiles/HISTORY.md); used by tools/
_MATH_X_BF16X3_32, _MATH_X_BF16X3_PIPE = 2, 4


def set_math_mode(mode: int) -> int:
    """Arithmetic of the GEMMs (include/lsnf_flow.h): MATH_BF16X3 (default: error-free three-way bf16 split, 24 operand
    bits, six bf16 MFMAs per product; 16x16x32 MFMA, the throughput forward software-pipelined where it applies),
    MATH_BF16X3_PHASED (the same without the pipelined forward), MATH_FP32 (fp32 MFMA) or MATH_FP16X2
    (opt-in, NARROWER than fp32: two-term fp16 split, three fp16 MFMAs per product, range-guarded by a bf16x3 fix-up pass).
    Returns the previous mode (mode < 0: query)."""
    return _lib.load().lsnf_set_math_mode(int(mode))


def new_act_saved(plan: "FlowPlan", B: int, device) -> torch.Tensor:
    """Uninitialised activation stash for `forward(..., act_saved=)` on a batch of B rows."""
    n = _lib.load().lsnf_act_saved_floats(plan.nz, plan.width, plan.depth, int(B))
    return torch.empty(max(n, 1), dtype=torch.float32, device=device)


def params_fast_path() -> bool:
    """True if the math mode in force lets `backward_params` run from the forward's stash (every bf16x3-family mode)."""
    return bool(_lib.load().lsnf_params_fast_path())


def new_params_workspace(plan: "FlowPlan", B: int, device) -> torch.Tensor:
    """Uninitialised workspace of `backward_params` for a batch of B rows (also the `params_ws` of `forward`)."""
    n = _lib.load().lsnf_backward_params_workspace_floats(plan.nz, plan.width, plan.depth, int(B))
    return torch.empty(max(n, 4), dtype=torch.float32, device=device)


STATS_DOUBLES = 264          # include/lsnf_flow.h LSNF_STATS_DOUBLES


def new_stats(device) -> torch.Tensor:
    """Zero-initialised accumulator (LSNF_STATS_DOUBLES = 264 doubles) for `forward(..., stats=)` (one per stream)."""
    return torch.zeros(STATS_DOUBLES, dtype=torch.float64, device=device)


def reverse(plan: FlowPlan, z: torch.Tensor, objective: Optional[torch.Tensor] = None):
    """model.py:484-498: returns (z_out, objective_out) with objective_out = objective - sum log|det J|."""
    lib = _lib.load()
    _need_cuda(z, "z")
    if z.dim() != 2 or z.shape[1] != plan.nz:
        raise LsnfError(f"z must be (B, {plan.nz}), got {tuple(z.shape)}")
    B = z.shape[0]
    if objective is not None:
        _need_cuda(objective, "objective")
    z_out = torch.empty_like(z)
    obj_out = torch.empty(B, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        rc = lib.lsnf_reverse(_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, B,
                              _ptr(z), _ptr(objective), _ptr(z_out), _ptr(obj_out), _stream_ptr(z.device))
    _lib.check(rc, "lsnf_reverse")
    return z_out, obj_out


def backward_z(plan: FlowPlan, z_out: torch.Tensor, z_saved: Optional[torch.Tensor],
               g_z1: Optional[torch.Tensor] = None, g_logdet: Optional[torch.Tensor] = None,
               ll_scale: Optional[float] = None, act_saved: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dz_in of the full stack (train.py:323).  Either pass upstream gradients (g_z1, g_logdet) or
    ll_scale for L = ll_scale * sum_b ll_b (train.py:320: ll_scale = -1).  act_saved: the stash the forward
    filled (same batch), or None: the stash is then rebuilt from the block outputs (`lsnf_restash`, bf16 matrix pipe;
    in MATH_FP32 the fp32 backward recomputes the coupling MLP itself)."""
    lib = _lib.load()
    _need_cuda(z_out, "z_out")
    B = z_out.shape[0]
    if act_saved is None and B > 0 and params_fast_path():
        # the rebuilt stash lives on the plan, one per (batch size, device, stream) -- ~1.4 KB per row, not re-allocated per call
        # (and not inside a graph capture); if the rebuild is refused, lsnf_backward_z recomputes the MLP itself (act_saved NULL)
        cache = plan.__dict__.setdefault("_restash_buffers", {})
        key = (B, z_out.device, torch.cuda.current_stream(z_out.device).cuda_stream)
        buf = cache.get(key)
        if buf is None:
            for k in [k for k in cache if k[2] == key[2]]:
                del cache[k]                 # one batch size at a time per stream
            buf = cache[key] = new_act_saved(plan, B, z_out.device)
        with torch.cuda.device(z_out.device):
            rc = lib.lsnf_restash(_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, B, _ptr(z_out), _ptr(z_saved),
                                  _ptr(buf), _stream_ptr(z_out.device))
        act_saved = buf if rc == 0 else None
    for name, t in (("z_saved", z_saved), ("g_z1", g_z1), ("g_logdet", g_logdet), ("act_saved", act_saved)):
        if t is not None:
            _need_cuda(t, name)
    g_in = torch.empty_like(z_out)
    with torch.cuda.device(z_out.device):
        rc = lib.lsnf_backward_z(_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, B,
                                 _ptr(z_out), _ptr(z_saved), _ptr(act_saved), _ptr(g_z1), _ptr(g_logdet),
                                 0 if ll_scale is None else 1, float(ll_scale or 0.0), _ptr(g_in),
                                 _stream_ptr(z_out.device))
    _lib.check(rc, "lsnf_backward_z")
    return g_in


class PhiloxNoise:
    """In-kernel N(0,1) noise for `langevin_step` (include/lsnf_flow.h `LsnfRng`): a pure function of
    (seed, offset, row0 + row, column).  `offset` must differ from step to step (`.step()` returns the next one);
    `row0` is the global index of the shard's first row, so that row-sharded chains draw what one GPU would;
    `offset_dev` is an optional int64/uint64 device scalar added to `offset` (advance it inside a captured graph)."""
    __slots__ = ("seed", "offset", "row0", "offset_dev")

    def __init__(self, seed: int, offset: int = 0, row0: int = 0, offset_dev: Optional[torch.Tensor] = None):
        self.seed, self.offset, self.row0, self.offset_dev = int(seed), int(offset), int(row0), offset_dev

    def step(self, n: int = 1) -> "PhiloxNoise":
        """A NEW generator n steps further on (this one is unchanged)."""
        return PhiloxNoise(self.seed, self.offset + n, self.row0, self.offset_dev)

    def advance(self, n: int = 1) -> "PhiloxNoise":
        """Moves THIS generator n steps on, in place (the K-step sampler calls it with K when it returns, so that a
        generator kept across training iterations never repeats a draw)."""
        self.offset += int(n)
        return self

    def _c(self):
        mask = (1 << 64) - 1
        dev = None if self.offset_dev is None else self.offset_dev.data_ptr()
        return _lib.LsnfRng(self.seed & mask, self.offset & mask, dev, self.row0)


def langevin_step(plan: FlowPlan, z: torch.Tensor, grad_g: Optional[torch.Tensor], noise,
                  step_size: float, *, inplace: bool = False, want_norms: bool = True, reuse_buffers: bool = False):
    """One flow-prior Langevin update (train.py:316-329) in two launches: forward (keeps block outputs) and the
    fused backward+update.  `noise`: None, a (B, nz) tensor of N(0,1) draws, or a `PhiloxNoise` (drawn inside the
    kernel).  Returns (z_new, ll, gf_norm, gg_norm); ll is the log-prob of the INPUT z
    (f_log_lkhd = -ll.sum(), train.py:320).  reuse_buffers: keep the intermediate buffers (block outputs, activation stash,
    z1, logdet, ll, norms: ~4 KB per row) on the plan between calls of the same batch size instead of asking the
    allocator for them every step -- the returned ll / norms are then overwritten by the next call (the K-step sampler
    consumes them at once)."""
    lib = _lib.load()
    _need_cuda(z, "z")
    B = z.shape[0]
    rng = None
    if isinstance(noise, PhiloxNoise):
        if noise.offset_dev is not None:
            od = noise.offset_dev
            if not od.is_cuda or od.dtype not in (torch.int64, torch.uint64) or od.numel() != 1:
                raise LsnfError("offset_dev must be one 64-bit integer on the GPU")
        rng, noise = ctypes.byref(noise._c()), None
    for name, t in (("grad_g", grad_g), ("noise", noise)):
        if t is not None:
            _need_cuda(t, name)
            if t.shape != z.shape:
                raise LsnfError(f"{name} must have the shape of z")
    cache = plan.__dict__.setdefault("_langevin_buffers", {}) if reuse_buffers else None
    key = (B, z.device, torch.cuda.current_stream(z.device).cuda_stream)     # (two streams sharing one plan must not share buffers)
    bufs = cache.get(key) if cache is not None else None
    if bufs is None:
        f32 = dict(dtype=torch.float32, device=z.device)
        bufs = {"act": new_act_saved(plan, B, z.device), "out": (torch.empty_like(z), torch.empty(B, **f32), torch.empty(B, **f32)),
                "saved": torch.empty((plan.depth - 1, B, plan.nz), **f32) if plan.depth > 1 else None,
                "gf": torch.empty(B, **f32), "gg": torch.empty(B, **f32)}
        if cache is not None:
            for k in [k for k in cache if k[2] == key[2]]:
                del cache[k]                 # one batch size at a time per stream
            cache[key] = bufs
    act, saved = bufs["act"], bufs["saved"]
    z1, logdet, ll, _ = forward(plan, z, None, want_ll=True, out=bufs["out"], act_saved=act, z_saved_out=saved)
    z_new = z if inplace else torch.empty_like(z)
    gf = bufs["gf"] if want_norms else None
    gg = bufs["gg"] if (want_norms and grad_g is not None) else None
    with torch.cuda.device(z.device):
        rc = lib.lsnf_langevin_step(_ptr(plan.buf), plan.nz, plan.width, plan.depth, plan.coupling, B,
                                    _ptr(z), _ptr(z1), _ptr(saved), _ptr(act), _ptr(grad_g), _ptr(noise), rng,
                                    float(step_size),
                                    _ptr(z_new), _ptr(gf), _ptr(gg), _stream_ptr(z.device))
    _lib.check(rc, "lsnf_langevin_step")
    return z_new, ll, gf, gg


def backward_params(plan: FlowPlan, params: Sequence[torch.Tensor], z_in: torch.Tensor, z_out: torch.Tensor,
                    z_saved: Optional[torch.Tensor], g_z1: Optional[torch.Tensor] = None,
                    g_logdet: Optional[torch.Tensor] = None, ll_scale: Optional[float] = None,
                    want_grad_z: bool = False, want_flat: bool = False, reuse_buffers: bool = False,
                    act_saved: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None):
    """dL/dtheta for the depth*12 live tensors (train.py:406-411).  Returns a list of gradients shaped like
    `params` (and dL/dz_in as a second value if want_grad_z; and, last, the ONE flat buffer the gradients are views
    of if want_flat -- a global norm / clip is then one reduction instead of 60).  reuse_buffers: keep the flat
    gradient buffer, its 60 views, the pointer tables and the workspace on the plan between calls (as long as the
    parameter storages and B do not change) -- the returned gradient tensors are then THE SAME objects every call,
    overwritten in place (an optimizer that consumes .grad before the next call does not notice; ~0.2 ms of host time).
    act_saved + workspace: the stash and the `params_ws` the forward of THIS evaluation was given -- the backward then runs
    from the stash on the bf16 matrix pipe instead of recomputing the coupling MLP in fp32 (`params_fast_path()`)."""
    lib = _lib.load()
    _need_cuda(z_in, "z_in")
    _need_cuda(z_out, "z_out")
    B = z_out.shape[0]
    if len(params) != plan.depth * LSNF_PARAMS_PER_BLOCK:
        raise LsnfError(f"expected {plan.depth * LSNF_PARAMS_PER_BLOCK} parameter tensors, got {len(params)}")
    for name, t in (("z_saved", z_saved), ("g_z1", g_z1), ("g_logdet", g_logdet), ("act_saved", act_saved), ("workspace", workspace)):
        if t is not None:
            _need_cuda(t, name)
    if (act_saved is None) != (workspace is None):
        raise LsnfError("act_saved and workspace go together: both from the forward of this evaluation")
    key = (tuple(p.data_ptr() for p in params), B, z_out.device, want_grad_z, torch.cuda.current_stream(z_out.device).cuda_stream)
    st = plan.__dict__.get("_bp_state") if reuse_buffers else None     # (keyed by stream too: a second stream re-allocates)
    if st is None or st["key"] != key:
        # one flat gradient buffer, handed out as views (60 separate allocations cost ~200 us of host time)
        raw = [p.detach() for p in params]
        for i, t in enumerate(raw):
            _need_cuda(t, f"param[{i}]")
        sizes = [t.numel() for t in raw]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=z_out.device)
        grads = [g.view(t.shape) for g, t in zip(flat.split(sizes), raw)]
        nws = lib.lsnf_backward_params_workspace_floats(plan.nz, plan.width, plan.depth, B) if workspace is None else 0
        base, o, offs = flat.data_ptr(), 0, []
        for n in sizes:
            offs.append(base + o * 4)
            o += n
        st = {"key": key, "flat": flat, "grads": grads,
              "g_in": torch.empty_like(z_out) if want_grad_z else None,
              "ws": torch.empty(nws, dtype=torch.float32, device=z_out.device),
              "parr": (ctypes.c_void_p * len(raw))(*key[0]), "garr": (ctypes.c_void_p * len(raw))(*offs)}
        if reuse_buffers:
            plan.__dict__["_bp_state"] = st
    flat, grads, g_in, ws, parr, garr = st["flat"], st["grads"], st["g_in"], st["ws"], st["parr"], st["garr"]
    if workspace is not None:
        need = lib.lsnf_backward_params_workspace_floats(plan.nz, plan.width, plan.depth, B)
        if workspace.numel() < need:
            raise LsnfError(f"workspace has {workspace.numel()} floats, lsnf_backward_params needs {need}")
        ws = workspace
    with torch.cuda.device(z_out.device):
        rc = lib.lsnf_backward_params(_ptr(plan.buf), parr, garr, plan.nz, plan.width, plan.depth, plan.coupling, B,
                                      _ptr(z_in), _ptr(z_out), _ptr(z_saved), _ptr(act_saved), _ptr(g_z1), _ptr(g_logdet),
                                      0 if ll_scale is None else 1, float(ll_scale or 0.0), _ptr(g_in), _ptr(ws),
                                      _stream_ptr(z_out.device))
    _lib.check(rc, "lsnf_backward_params")
    out = (grads,) + ((g_in,) if want_grad_z else ()) + ((flat,) if want_flat else ())
    return out if len(out) > 1 else grads
