"""CPU oracle for the latent flow prior -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  The shipped path (package
``latent-space-normalizing-flow_amd`` -> ``liblsnf_flow.so``) never routes through it.

It restates, op for op and in the same evaluation order, the algorithm of the
reference's flow prior using stock PyTorch-CPU tensor ops (the reference itself is
PyTorch eager code, so this is the closest possible "port"):

  reference /root/reference/model.py
    actnorm                     :227-294
    invertible_1x1_conv         :171-198
    fc / fc_zeros / f           :296-350
    revnet2d_step (one block)   :367-458
    revnet2d                    :352-365
    _netF.forward               :473-498
  reference /root/reference/train.py
    log-prob assembly           :316-320, 406-410
    Langevin update             :311-326

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference's own ``model.py`` in the build container (``tests/golden/make_golden.py``)
and ``tests/test_oracle_golden.py`` checks every function below against them.

All functions are functional: parameters come in as a ``dict`` keyed exactly like the
reference's ``_netF.state_dict()``, e.g.
``revnet2d_s.0.revnet2d_step_s.3.f.fc_zeros.logs``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LOG_2PI = float(np.log(2 * np.pi))  # train.py:318 uses np.log(2*np.pi)


def block_prefix(i: int, level: int = 0) -> str:
    return f"revnet2d_s.{level}.revnet2d_step_s.{i}."


def depth_of(params: Params) -> int:
    d = 0
    while block_prefix(d) + "actnorm.logs" in params:
        d += 1
    return d


# ----------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------
def actnorm_fwd(x: Tensor, b: Tensor, logs: Tensor, logdet: Tensor | None = None):
    """model.py:243-244 (center), :264-276 (scale); logscale_factor=3 (model.py:250)."""
    x = x + b                                   # model.py:244
    ls = logs * 3.0                             # model.py:264
    x = x * torch.exp(ls)                       # model.py:268
    if logdet is not None:
        return x, logdet + torch.sum(ls)        # model.py:273-276
    return x


def actnorm_rev(x: Tensor, b: Tensor, logs: Tensor, logdet: Tensor | None = None):
    """model.py:288-291: scale^-1 first (:270), then un-center (:246)."""
    ls = logs * 3.0
    x = x * torch.exp(-ls)                      # model.py:270
    if logdet is not None:
        logdet = logdet + (-1.0) * torch.sum(ls)  # model.py:273-276 with reverse
    x = x - b                                   # model.py:246
    if logdet is not None:
        return x, logdet
    return x


def conv1x1_logabsdet(w: Tensor) -> Tensor:
    """model.py:182 -- determinant in float64, result ALWAYS cast to float32 (the reference
    writes `.float()`, so even a double-precision module rounds this term to fp32)."""
    return torch.log(torch.abs(torch.det(w.double()))).float()


def mlp_f(p: Params, pre: str, z1: Tensor) -> Tensor:
    """model.py:306-310 with fc (:321-332, actnorm branch) and fc_zeros (:344-350)."""
    h = torch.matmul(z1, p[pre + "f.fc_1.w"])                                  # :326
    h = actnorm_fwd(h, p[pre + "f.fc_1.actnorm.b"], p[pre + "f.fc_1.actnorm.logs"])  # :328
    h = torch.relu(h)                                                          # :307
    h = torch.matmul(h, p[pre + "f.fc_2.w"])
    h = actnorm_fwd(h, p[pre + "f.fc_2.actnorm.b"], p[pre + "f.fc_2.actnorm.logs"])
    h = torch.relu(h)                                                          # :308
    h = torch.matmul(h, p[pre + "f.fc_zeros.w"])                               # :347
    h = h + p[pre + "f.fc_zeros.b"]                                            # :348
    h = h * torch.exp(p[pre + "f.fc_zeros.logs"] * 3.0)                        # :349
    return h


def coupling_of(p: Params) -> int:
    """1 = affine (fc_zeros has nz outputs, model.py:387), 0 = additive (nz/2 outputs, model.py:385)."""
    pre = block_prefix(0)
    return 1 if p[pre + "f.fc_zeros.w"].shape[1] == p[pre + "actnorm.logs"].shape[1] else 0


def block_fwd(p: Params, i: int, z: Tensor, logdet: Tensor, coupling: int | None = None):
    """One revnet2d_step, forward branch: model.py:391-422 (permutation 2)."""
    coupling = coupling_of(p) if coupling is None else coupling
    pre = block_prefix(i)
    n_z = z.shape[-1]
    z, logdet = actnorm_fwd(z, p[pre + "actnorm.b"], p[pre + "actnorm.logs"], logdet)  # :392
    w = p[pre + "invertible_1x1_conv.w"]
    dlogdet = conv1x1_logabsdet(w)                                              # :182
    z = torch.matmul(z, w)                                                      # :187
    logdet = logdet + dlogdet                                                   # :189
    z1 = z[:, : n_z // 2]                                                       # :404
    z2 = z[:, n_z // 2:]                                                        # :405
    if coupling == 0:
        z2 = z2 + mlp_f(p, pre, z1)                                             # :408
    else:
        h = mlp_f(p, pre, z1)                                                   # :410
        shift = h[:, 0::2]                                                      # :411
        scale = torch.sigmoid(h[:, 1::2] + 2.0)                                 # :413
        z2 = z2 + shift                                                         # :414
        z2 = z2 * scale                                                         # :415
        logdet = logdet + torch.sum(torch.log(scale), dim=1)                    # :418
    z = torch.cat([z1, z2], 1)                                                  # :422
    return z, logdet


def block_rev(p: Params, i: int, z: Tensor, logdet: Tensor, coupling: int | None = None):
    """One revnet2d_step, reverse branch: model.py:424-456.  Functional (the reference
    mutates its inputs in place, model.py:436-438; values are identical)."""
    coupling = coupling_of(p) if coupling is None else coupling
    pre = block_prefix(i)
    n_z = z.shape[-1]
    z1 = z[:, : n_z // 2]
    z2 = z[:, n_z // 2:]
    if coupling == 0:
        z2 = z2 - mlp_f(p, pre, z1)                                             # :430
    else:
        h = mlp_f(p, pre, z1)
        shift = h[:, 0::2]
        scale = torch.sigmoid(h[:, 1::2] + 2.0)
        z2 = z2 / scale                                                         # :436
        z2 = z2 - shift                                                         # :437
        logdet = logdet - torch.sum(torch.log(scale), dim=1)                    # :438
    z = torch.cat([z1, z2], 1)                                                  # :445
    w = p[pre + "invertible_1x1_conv.w"]
    dlogdet = conv1x1_logabsdet(w)
    z = torch.matmul(z, torch.inverse(w))                                       # :193-194
    logdet = logdet - dlogdet                                                   # :196
    z, logdet = actnorm_rev(z, p[pre + "actnorm.b"], p[pre + "actnorm.logs"], logdet)  # :456
    return z, logdet


# ----------------------------------------------------------------------------------
# the stack (= _netF.forward for f_n_levels == 1)
# ----------------------------------------------------------------------------------
def flow_forward(p: Params, z: Tensor, objective: Tensor, coupling: int | None = None,
                 first_block: int = 0, n_blocks: int | None = None):
    """_netF.forward(reverse=False): model.py:474-483 -> revnet2d :358-360."""
    d = depth_of(p)
    n_blocks = d - first_block if n_blocks is None else n_blocks
    for i in range(first_block, first_block + n_blocks):
        z, objective = block_fwd(p, i, z, objective, coupling)
    return z, objective


def flow_reverse(p: Params, z: Tensor, objective: Tensor, coupling: int | None = None):
    """_netF.forward(reverse=True, return_obj=True): model.py:485-498.
    Returns (z, -objective) exactly like the reference does with return_obj=True."""
    d = depth_of(p)
    for i in reversed(range(d)):
        z, objective = block_rev(p, i, z, objective, coupling)
    return z, -objective


def log_prob(z1: Tensor, logdet: Tensor) -> Tensor:
    """train.py:317-319.  NB the constant is +log(2*pi), once, as the reference writes it."""
    prior_ll = -0.5 * (z1 ** 2)
    prior_ll = prior_ll.flatten(1).sum(-1) + LOG_2PI
    return prior_ll + logdet


def flow_log_prob(p: Params, z: Tensor, coupling: int | None = None):
    z1, logdet = flow_forward(p, z, torch.zeros(z.shape[0], dtype=z.dtype), coupling)
    return z1, logdet, log_prob(z1, logdet)


def grad_neg_sum_ll_wrt_z(p: Params, z: Tensor, coupling: int | None = None) -> Tensor:
    """train.py:316-323: d(-sum_b ll)/dz by autograd over the restated ops."""
    zz = z.clone().detach().requires_grad_(True)
    _, _, ll = flow_log_prob(p, zz, coupling)
    (g,) = torch.autograd.grad(-ll.sum(), zz)
    return g


def grad_neg_mean_ll_wrt_params(p: Params, z: Tensor, coupling: int | None = None) -> Dict[str, Tensor]:
    """train.py:406-411: d(-mean_b ll)/dtheta for every parameter that receives a gradient
    (the fc_*.b tensors never do; the duplicate 'actnorm.bias' key aliases 'actnorm.b')."""
    live = {k: v.clone().detach().requires_grad_(True) for k, v in p.items() if is_live_param(k)}
    q = dict(p)
    q.update(live)
    _, _, ll = flow_log_prob(q, z, coupling)
    keys = sorted(live)
    grads = torch.autograd.grad(-ll.mean(), [live[k] for k in keys])
    return dict(zip(keys, grads))


def relu_margin(p: Params, z: Tensor) -> Tensor:
    """(B,) float64: the smallest |pre-activation| any ReLU of the stack sees for each row, in
    float64.  d ll/dz is DISCONTINUOUS where a pre-activation crosses zero (model.py:307-308), so
    for a row whose margin is below fp32 rounding noise (~1e-6) two correct fp32 implementations may
    legitimately return gradients that differ by O(1e-3).  Tests use this to set such rows aside."""
    p64 = to_dtype(p, torch.float64)
    zz = z.double()
    ld = torch.zeros(zz.shape[0], dtype=torch.float64)
    margin = torch.full((zz.shape[0],), float("inf"), dtype=torch.float64)
    for i in range(depth_of(p64)):
        pre = block_prefix(i)
        n_z = zz.shape[-1]
        x = actnorm_fwd(zz, p64[pre + "actnorm.b"], p64[pre + "actnorm.logs"])
        v1 = torch.matmul(x, p64[pre + "invertible_1x1_conv.w"])[:, : n_z // 2]
        a1 = actnorm_fwd(torch.matmul(v1, p64[pre + "f.fc_1.w"]), p64[pre + "f.fc_1.actnorm.b"], p64[pre + "f.fc_1.actnorm.logs"])
        a2 = actnorm_fwd(torch.matmul(torch.relu(a1), p64[pre + "f.fc_2.w"]), p64[pre + "f.fc_2.actnorm.b"],
                         p64[pre + "f.fc_2.actnorm.logs"])
        margin = torch.minimum(margin, torch.minimum(a1.abs().min(1).values, a2.abs().min(1).values))
        zz, ld = block_fwd(p64, i, zz, ld)
    return margin


def is_live_param(key: str) -> bool:
    """Parameters that get a gradient in the reference: everything except
    fc_1.b / fc_2.b (unused, model.py:319,327-330) and the '.bias' alias (model.py:231)."""
    if key.endswith(".bias"):
        return False
    if key.endswith("f.fc_1.b") or key.endswith("f.fc_2.b"):
        return False
    return True


def langevin_prior_step(p: Params, z: Tensor, grad_g: Tensor | None, step_size: float,
                        noise: Tensor | None = None):
    """train.py:316-326 restricted to the flow-prior part: given the generator's
    gradient (or None -> 0) do  z <- z - 0.5 s^2 (grad_g + grad_f) [+ s * noise].
    Returns (z_new, f_log_lkhd, grad_f)."""
    zz = z.clone().detach().requires_grad_(True)
    _, _, ll = flow_log_prob(p, zz)
    f = -ll.sum()                                            # train.py:320
    (gf,) = torch.autograd.grad(f, zz)                       # train.py:323
    g = gf if grad_g is None else grad_g + gf
    z_new = z - 0.5 * step_size * step_size * g              # train.py:324
    if noise is not None:
        z_new = z_new + step_size * noise                    # train.py:326
    return z_new.detach(), f.detach(), gf.detach()


# ----------------------------------------------------------------------------------
# reference-style initialisation (model.py:176, 230-233, 318-319, 340-342)
# ----------------------------------------------------------------------------------
def init_params(nz: int, width: int, depth: int = 5, seed: int = 1, fcz_std: float = 0.05,
                all_std: float = 0.0, dtype=torch.float32) -> Params:
    """Synthetic weights with the reference's initial distributions (not its RNG stream):
    actnorm b/logs ~ 0.05 N(0,1); conv w = random orthogonal (QR); fc w ~ 0.05 N(0,1);
    fc_zeros = 0 + fcz_std N(0,1) (perturbed, SURVEY 8a: else the coupling is trivial);
    all_std adds extra N(0,1)*all_std noise on every tensor ("trained-like")."""
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    half = nz // 2
    p: Params = {}

    def rn(*shape, std):
        return torch.randn(*shape, generator=g, dtype=torch.float64) * std

    for i in range(depth):
        pre = block_prefix(i)
        p[pre + "actnorm.b"] = rn(1, nz, std=0.05)
        p[pre + "actnorm.logs"] = rn(1, nz, std=0.05)
        q = np.linalg.qr(rs.randn(nz, nz))[0]
        p[pre + "invertible_1x1_conv.w"] = torch.tensor(q, dtype=torch.float64)
        for name, n_in in (("fc_1", half), ("fc_2", width)):
            p[pre + f"f.{name}.w"] = rn(n_in, width, std=0.05)
            p[pre + f"f.{name}.b"] = torch.zeros(1, width, dtype=torch.float64)
            p[pre + f"f.{name}.actnorm.b"] = rn(1, width, std=0.05)
            p[pre + f"f.{name}.actnorm.logs"] = rn(1, width, std=0.05)
        p[pre + "f.fc_zeros.w"] = rn(width, nz, std=fcz_std)
        p[pre + "f.fc_zeros.b"] = rn(1, nz, std=fcz_std)
        p[pre + "f.fc_zeros.logs"] = rn(1, nz, std=fcz_std)
    if all_std > 0:
        for k in list(p):
            if is_live_param(k):
                p[k] = p[k] + rn(*p[k].shape, std=all_std)
    out = {k: v.to(dtype).contiguous() for k, v in p.items()}
    for i in range(depth):                      # the alias the reference registers (model.py:231)
        pre = block_prefix(i)
        out[pre + "actnorm.bias"] = out[pre + "actnorm.b"]
        out[pre + "f.fc_1.actnorm.bias"] = out[pre + "f.fc_1.actnorm.b"]
        out[pre + "f.fc_2.actnorm.bias"] = out[pre + "f.fc_2.actnorm.b"]
    return out


def to_dtype(p: Params, dtype) -> Params:
    return {k: v.to(dtype) for k, v in p.items()}


def state_dict_keys(depth: int = 5) -> List[str]:
    """The 17 keys per block of the reference's state_dict, in its registration order."""
    per_block = [
        "actnorm.b", "actnorm.bias", "actnorm.logs", "invertible_1x1_conv.w",
        "f.fc_1.w", "f.fc_1.b", "f.fc_1.actnorm.b", "f.fc_1.actnorm.bias", "f.fc_1.actnorm.logs",
        "f.fc_2.w", "f.fc_2.b", "f.fc_2.actnorm.b", "f.fc_2.actnorm.bias", "f.fc_2.actnorm.logs",
        "f.fc_zeros.w", "f.fc_zeros.b", "f.fc_zeros.logs",
    ]
    return [block_prefix(i) + k for i in range(depth) for k in per_block]
