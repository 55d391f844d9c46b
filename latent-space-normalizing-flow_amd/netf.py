"""`_netF`: drop-in replacement for the reference's flow prior (reference model.py:460-498).

Same constructor (`_netF(hps, nz)`), same `forward` signature and return conventions, the same
85-key `state_dict` (so reference checkpoints load unchanged, train.py:343-348), the same initial
distributions drawn in the same RNG order -- but every number is produced by the HIP kernels of
liblsnf_flow.so.  The sub-modules below only hold parameters under the reference's names; the
compute path is a single `torch.autograd.Function` around the fused stack kernels.

There is no CPU path: calling the module on CPU tensors raises `LsnfError`.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import flow
from ._lib import LsnfError


# ---------------------------------------------------------------------------------------------
# parameter holders (names / shapes / init mirror reference model.py:171-177, 227-233, 312-319,
# 334-342, 296-303, 367-387, 352-356)
# ---------------------------------------------------------------------------------------------
class actnorm(nn.Module):
    def __init__(self, nz):
        super().__init__()
        self.b = nn.Parameter(torch.randn(1, nz) * 0.05)
        self.register_parameter(name="bias", param=self.b)      # the reference's duplicate key (model.py:231)
        self.logs = nn.Parameter(torch.randn(1, nz) * 0.05)


class invertible_1x1_conv(nn.Module):
    def __init__(self, width, nz):
        super().__init__()
        w_init = np.linalg.qr(np.random.randn(nz, nz))[0].astype("float32")   # random orthogonal (model.py:176)
        self.w = nn.Parameter(torch.tensor(w_init, dtype=torch.float))


class fc(nn.Module):
    def __init__(self, n_in, width):
        super().__init__()
        self.width = width
        self.actnorm = actnorm(nz=width)
        self.w = nn.Parameter(torch.randn(n_in, width) * 0.05)
        self.b = nn.Parameter(torch.zeros(1, width))            # unused by the reference's forward (model.py:327-330)


class fc_zeros(nn.Module):
    def __init__(self, n_in, width):
        super().__init__()
        self.width = width
        self.w = nn.Parameter(torch.zeros(n_in, width))
        self.b = nn.Parameter(torch.zeros(1, width))
        self.logs = nn.Parameter(torch.zeros(1, width))


class f(nn.Module):
    def __init__(self, width, n_in=None, n_out=None):
        super().__init__()
        self.n_out = n_out
        self.fc_1 = fc(n_in, width)
        self.fc_2 = fc(width, width)
        self.fc_zeros = fc_zeros(width, n_out)


class revnet2d_step(nn.Module):
    def __init__(self, id, hps, nz):
        super().__init__()
        self.actnorm = actnorm(nz=nz)
        if hps.f_flow_permutation == 2:
            self.invertible_1x1_conv = invertible_1x1_conv(hps.f_width, nz=nz)
            self.shuffle_features = None
        else:
            # the reference's permutations 0 and 1 do not run; only the 1x1 conv exists here
            raise Exception("only f_flow_permutation == 2 (invertible 1x1 conv) is supported")
        self.id = id
        assert nz % 2 == 0
        if hps.f_flow_coupling == 1:
            self.f = f(hps.f_width, nz // 2, nz)
        elif hps.f_flow_coupling == 0:
            self.f = f(hps.f_width, nz // 2, nz // 2)      # additive coupling: shift only (model.py:385,407-408)
        else:
            raise Exception()

    def live_parameters(self) -> List[nn.Parameter]:
        """The 12 tensors that reach the kernels, in the ABI's order (include/lsnf_flow.h)."""
        ff = self.f
        return [self.actnorm.b, self.actnorm.logs, self.invertible_1x1_conv.w,
                ff.fc_1.w, ff.fc_1.actnorm.b, ff.fc_1.actnorm.logs,
                ff.fc_2.w, ff.fc_2.actnorm.b, ff.fc_2.actnorm.logs,
                ff.fc_zeros.w, ff.fc_zeros.b, ff.fc_zeros.logs]


class revnet2d(nn.Module):
    def __init__(self, hps, nz):
        super().__init__()
        self.hps = hps
        self.revnet2d_step_s = nn.ModuleList([revnet2d_step(str(i), hps, nz=nz) for i in range(hps.f_depth)])


# ---------------------------------------------------------------------------------------------
# autograd bridge
# ---------------------------------------------------------------------------------------------
class _Upstream:
    """Hand-over between the two autograd nodes below (filled by _FlowStackFn.backward)."""
    __slots__ = ("g_z1", "g_logdet", "z", "z1", "saved", "plan_key", "act", "ws")

    def __init__(self):
        self.g_z1 = self.g_logdet = self.z = self.z1 = self.saved = self.plan_key = self.act = self.ws = None


class _ParamGate(torch.autograd.Function):
    """(*live_params) -> scalar token.  Exists so that parameter gradients are computed ONLY when the
    autograd engine actually needs them: `torch.autograd.grad(f, z)` in the Langevin loop (train.py:323)
    never executes this node's backward, `loss_f.backward()` (train.py:411) does."""

    @staticmethod
    def forward(ctx, module, holder, *params):
        ctx.module, ctx.holder, ctx.n = module, holder, len(params)
        return params[0].new_zeros(())

    @staticmethod
    def backward(ctx, _g_token):
        h, module = ctx.holder, ctx.module
        if h.z1 is None:
            raise LsnfError("parameter gradients requested before the flow's backward ran")
        if module._current_key() != h.plan_key:     # the LIVE parameters, not the key cached at the last _plan() call
            raise LsnfError("flow parameters were modified between forward and backward")
        grads = flow.backward_params(module._plan(), module._param_list(), h.z, h.z1, h.saved, h.g_z1, h.g_logdet,
                                     act_saved=h.act if h.ws is not None else None, workspace=h.ws)
        grads = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[2:])]
        return (None, None, *grads)


class _FlowStackFn(torch.autograd.Function):
    """(z, objective, token) -> (z1, logdet).  Forward: lsnf_forward (block outputs kept for the backward);
    backward: lsnf_backward_z for z; the upstream gradients are handed to _ParamGate for the parameters."""

    @staticmethod
    def forward(ctx, module, holder, z, objective, token):
        plan = module._plan()
        # z needs a gradient (the Langevin sampler, train.py:316-323): also keep the sigmoid / relu-mask stash so
        # that lsnf_backward_z does not recompute the coupling MLP.  z is a constant and the parameters are the leaves
        # (the flow-MLE step, train.py:404-411): keep the stash AND let the forward write the hidden activations into
        # the parameter-gradient workspace, so that lsnf_backward_params runs from the stash (flow.params_fast_path).
        B = z.shape[0]
        for_params = bool(B) and not ctx.needs_input_grad[2] and ctx.needs_input_grad[4] and flow.params_fast_path()
        act = flow.new_act_saved(plan, B, z.device) if (B and (ctx.needs_input_grad[2] or for_params)) else None
        ws = flow.new_params_workspace(plan, B, z.device) if for_params else None
        z1, logdet, _, saved = flow.forward(plan, z, objective, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
        ctx.module, ctx.holder = module, holder
        ctx.plan_key = module._plan_key
        ctx.act, ctx.ws = act, ws
        ctx.save_for_backward(z, z1, saved if saved is not None else z1.new_empty(0))
        return z1, logdet

    @staticmethod
    def backward(ctx, g_z1, g_logdet):
        module, h = ctx.module, ctx.holder
        z, z1, saved = ctx.saved_tensors
        if module._current_key() != ctx.plan_key:   # the LIVE parameters: an optimizer step / load_state_dict / EMA copy
            # between forward and backward would otherwise re-prepare the plan in place and pair new weights with the
            # activations saved from the old ones (PyTorch autograd raises a version-counter error in the same situation)
            raise LsnfError("flow parameters were modified between forward and backward")
        saved_t = saved if saved.numel() else None
        g_z1 = None if g_z1 is None else g_z1.contiguous()
        g_logdet = None if g_logdet is None else g_logdet.contiguous()
        g_z = flow.backward_z(module._plan(), z1, saved_t, g_z1, g_logdet, act_saved=ctx.act) if ctx.needs_input_grad[2] else None
        g_obj = g_logdet if ctx.needs_input_grad[3] else None
        g_tok = None
        if ctx.needs_input_grad[4]:
            h.g_z1, h.g_logdet, h.z, h.z1, h.saved, h.plan_key = g_z1, g_logdet, z, z1, saved_t, ctx.plan_key
            h.act, h.ws = ctx.act, ctx.ws
            g_tok = z1.new_zeros(())
        return None, None, g_z, g_obj, g_tok


class _netF(nn.Module):
    """Reference model.py:460-498.  `hps` needs f_n_levels, f_depth, f_flow_permutation, f_width,
    f_flow_coupling (model.py:355-356,372-387,465)."""

    def __init__(self, hps, nz, *args, **kwargs):
        super().__init__(*args, **kwargs)
        revnet2d_s = []
        self.hps = hps
        self.nz = nz
        for i in range(hps.f_n_levels):
            revnet2d_s.append(revnet2d(hps, nz=nz))
            if i < hps.f_n_levels - 1:
                raise NotImplementedError      # as the reference (model.py:467-470): no split layer
        self.revnet2d_s = nn.ModuleList(revnet2d_s)
        self._cached_plan: Optional[flow.FlowPlan] = None
        self._plan_key = None

    # ---- prepared weights, re-derived only when a parameter changed -------------------------------
    def _param_list(self) -> List[nn.Parameter]:
        """The depth*12 live Parameter objects in ABI order (cached: module traversal costs ~80 us per call;
        `.to()`, optimizer steps and load_state_dict keep the Parameter objects, only their storage changes)."""
        cached = self.__dict__.get("_live_params")
        if cached is None:
            cached = []
            for step in self.revnet2d_s[0].revnet2d_step_s:
                cached += step.live_parameters()
            self.__dict__["_live_params"] = cached
        return cached

    def _apply(self, fn, *args, **kwargs):     # .to() / .cuda() / .float(): drop caches that hold device state
        self.__dict__.pop("_live_params", None)
        self._cached_plan, self._plan_key = None, None
        return super()._apply(fn, *args, **kwargs)

    def _current_key(self):
        """Identity of the live parameter values as autograd sees them: (storage address, version counter) per tensor.
        In-place updates through the Parameter (optimizer steps, `load_state_dict`, `p.mul_()` under no_grad) bump the
        version; writes through `p.data` (`p.data.copy_()`, `p.data.clamp_()`) do NOT -- `.data` has its own counter --
        so code that edits weights that way must call `invalidate_plan()`."""
        return tuple((p.data_ptr(), p._version) for p in self._param_list())

    def invalidate_plan(self) -> None:
        """Force the next call to re-derive the prepared weights (after writes through `p.data`, which no version
        counter records)."""
        self._plan_key = None

    @staticmethod
    def _require_gpu(params) -> None:
        for p in params:
            if not p.is_cuda:
                raise LsnfError("_netF lives on %s: move it to the GPU (netF.to(device)); there is no CPU path" % p.device)

    def _plan(self) -> flow.FlowPlan:
        params = self._param_list()
        key = self._current_key()
        if self._cached_plan is None or key != self._plan_key or self._cached_plan.device != params[0].device:
            self._require_gpu(params)
            reuse = self._cached_plan if (self._cached_plan is not None and
                                          self._cached_plan.device == params[0].device) else None
            with torch.no_grad():
                tensors = [p.detach().contiguous() for p in params]
                self._cached_plan = flow.prepare(tensors, self.nz, self.hps.f_width, self.hps.f_depth,
                                                 self.hps.f_flow_coupling, plan=reuse)
            self._plan_key = key
        return self._cached_plan

    # ---- reference forward signature (model.py:473) ---------------------------------------------
    def forward(self, z, objective, init=False, reverse=False, eps=None, eps_std=None, z2_s=None, return_obj=False):
        if init:
            raise NotImplementedError("data-dependent actnorm init (init=True) is never used by the reference's "
                                      "train.py (all call sites pass init=False) and is not implemented")
        if z.dim() != 2:
            raise ValueError(f"z must be (B, nz); got shape {tuple(z.shape)} (note: the reference's "
                             f"torch.squeeze(z) call site breaks for B == 1, train.py:316)")
        z = z.contiguous()
        objective = objective.contiguous()
        if not reverse:
            params = self._param_list()
            if torch.is_grad_enabled() and (z.requires_grad or objective.requires_grad or
                                            any(p.requires_grad for p in params)):
                holder = _Upstream()
                token = _ParamGate.apply(self, holder, *params)
                z1, logdet = _FlowStackFn.apply(self, holder, z, objective, token)
            else:
                z1, logdet, _, _ = flow.forward(self._plan(), z.detach(), objective.detach(), want_ll=False)
            return z1, logdet, []
        if torch.is_grad_enabled() and z.requires_grad:
            raise NotImplementedError("reverse pass is inference-only (the reference calls it under no_grad, "
                                      "train.py:433-434,473-475,569-571)")
        x, obj = flow.reverse(self._plan(), z.detach(), objective.detach())
        if not return_obj:
            return x
        return x, -obj

    # ---- fused extras (not in the reference; train.py:316-323 collapsed into two launches) ---------
    def log_prob(self, z, stats=None):
        """(z1, logdet, ll) with ll = -0.5*sum z1^2 + log(2*pi) + logdet (train.py:317-319), one launch.
        stats: optional buffer from `flow.new_stats()` (or a `PipelinedStatsReducer` row): the launch also leaves
        [sum ll, sum logdet, rows] in stats[4:7] (train.py:320 -- what a row-sharded evaluation all-reduces)."""
        z1, logdet, ll, _ = flow.forward(self._plan(), z.detach().contiguous(), None, want_ll=True, stats=stats)
        return z1, logdet, ll

    def log_prob_and_grad(self, z, scale=-1.0):
        """ll and d(scale * sum ll)/dz (train.py:320-323 uses scale = -1), two launches."""
        plan = self._plan()
        act = flow.new_act_saved(plan, z.shape[0], z.device)
        z1, logdet, ll, saved = flow.forward(plan, z.detach().contiguous(), None, want_ll=True, save_for_backward=True,
                                             act_saved=act)
        g = flow.backward_z(plan, z1, saved, ll_scale=scale, act_saved=act)
        return ll, g

    def langevin_step(self, z, grad_g=None, noise=None, step_size=0.1, inplace=False, reuse_buffers=False):
        """z <- z - 0.5 s^2 (grad_g + d(-sum ll)/dz) + s*noise (train.py:316-326), flow part fused into two launches.
        noise: None, a tensor of N(0,1) draws, or a `flow.PhiloxNoise` (drawn inside the kernel).
        Returns (z_new, ll_of_input_z, |grad_f| per row, |grad_g| per row or None)."""
        if isinstance(noise, torch.Tensor):
            noise = noise.detach().contiguous()
        return flow.langevin_step(self._plan(), z.detach().contiguous(),
                                  None if grad_g is None else grad_g.detach().contiguous(), noise, step_size,
                                  inplace=inplace, reuse_buffers=reuse_buffers)

    def mle_grads(self, z, accumulate: bool = False, max_norm: Optional[float] = None, reuse_buffers: bool = False):
        """Fused flow-MLE gradients (train.py:404-411): loss_f = -mean_b ll(z_b) and d loss_f / d theta written to
        `.grad` of the 60 live tensors -- forward (ll summed in-kernel), dump backward, batch contraction, unfold:
        5 launches and no autograd graph.  max_norm: clip the global gradient norm (train.py:413-414
        `clip_grad_norm_`) on the one flat buffer the gradients are views of (3 launches instead of a foreach over
        60 tensors; only without `accumulate`).  reuse_buffers: the gradient tensors are the same objects every call,
        overwritten in place (`flow.backward_params`): for loops that consume `.grad` before the next call.
        Returns loss_f as a 0-dim device tensor (no host sync)."""
        plan = self._plan()
        z = z.detach().contiguous()
        B = z.shape[0]
        if B == 0:
            raise LsnfError("mle_grads needs a non-empty batch")
        stats = flow.new_stats(z.device)
        fast = flow.params_fast_path()         # forward keeps the stash + writes h1 / h2 for the contraction: nothing is recomputed
        bufs = plan.__dict__.get("_mle_buffers") if reuse_buffers else None
        bkey = (B, z.device, torch.cuda.current_stream(z.device).cuda_stream)
        if fast and (bufs is None or bufs[0] != bkey):
            bufs = (bkey, flow.new_act_saved(plan, B, z.device), flow.new_params_workspace(plan, B, z.device))
            if reuse_buffers:
                plan.__dict__["_mle_buffers"] = bufs
        act, ws = (bufs[1], bufs[2]) if fast else (None, None)
        z1, _, _, saved = flow.forward(plan, z, None, want_ll=False, save_for_backward=True, stats=stats, act_saved=act, params_ws=ws)
        params = self._param_list()
        grads, flat = flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, want_flat=True,
                                           reuse_buffers=reuse_buffers and not accumulate, act_saved=act, workspace=ws)
        if max_norm is not None:
            if accumulate:
                raise LsnfError("max_norm clips the gradients of this call only: not with accumulate=True")
            flat.mul_(torch.clamp(float(max_norm) / (flat.norm() + 1e-6), max=1.0))   # torch's clip_grad_norm_ formula
        for p, g in zip(params, grads):
            if not p.requires_grad or p.grad is g:
                continue
            p.grad = g if (p.grad is None or not accumulate) else p.grad + g
        return (stats[4] * (-1.0 / B)).to(torch.float32)
