"""Caller-side harness around the flow prior: the reference's short-run Langevin sampler and flow-MLE step,
restated (they are closures inside train.py and cannot be imported) on top of the fused kernels.

  sample_langevin_post_z_with_flow  <- train.py:307-335 (training) / :602-634 (testing: 20x steps, no noise)
  flow_mle_step                     <- train.py:404-415

The generator `netG` is any `nn.Module` mapping (B, nz, 1, 1) -> images (the reference's `_netG`, stock PyTorch /
MIOpen, is out of scope of this build and used as is); its z-gradient comes from torch autograd exactly as in the
reference, the flow's comes from `lsnf_langevin_step` (two launches per step instead of ~750 eager kernels)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn


def sample_langevin_post_z_with_flow(z, x, netG: nn.Module, netF, *, g_l_steps: int, g_l_step_size: float,
                                     g_llhd_sigma: float, g_l_with_noise: bool = True,
                                     generator: Optional[torch.Generator] = None, philox=None):
    """Returns (z_k detached (B,nz,1,1), mean |z_grad_g|, mean |z_grad_f|, f_log_lkhd of the last step's input).
    Noise (train.py:325-326): `torch.randn` draws (optionally from `generator`), or, with `philox` =
    `flow.PhiloxNoise(seed, offset, row0)`, drawn inside the update kernel (step k uses offset + k): no randn
    launch, no (B, nz) noise tensor, and the same draws however the rows are sharded over GPUs.  On return `philox` has
    been ADVANCED by g_l_steps (in place): a generator object kept across training iterations continues its stream
    instead of repeating it."""
    z = z.clone().detach()
    B, nz = z.shape[0], z.shape[1]
    mse = nn.MSELoss(reduction="sum")
    gg_norm = gf_norm = f_log_lkhd = None
    for k in range(g_l_steps):
        z.requires_grad_(True)
        x_hat = netG(z)                                                                      # train.py:312
        g_log_lkhd = 1.0 / (2.0 * g_llhd_sigma * g_llhd_sigma) * mse(x_hat, x)               # train.py:313
        z_grad_g = torch.autograd.grad(g_log_lkhd, z)[0]                                     # train.py:314
        z2d = z.detach().view(B, nz)
        noise = None
        if g_l_with_noise:                                                                   # train.py:325-326
            noise = philox.step(k) if philox is not None else \
                torch.randn(z2d.shape, device=z2d.device, dtype=z2d.dtype, generator=generator)
        z_new, ll, gf, gg = netF.langevin_step(z2d, z_grad_g.reshape(B, nz), noise, g_l_step_size,    # :316-326
                                               reuse_buffers=True)
        f_log_lkhd = -ll.sum()                                                               # train.py:320
        gg_norm, gf_norm = gg.mean(), gf.mean()                                              # train.py:328-329
        z = z_new.view(B, nz, 1, 1)
    if philox is not None and g_l_with_noise:
        philox.advance(g_l_steps)
    return z.detach(), gg_norm, gf_norm, f_log_lkhd


def flow_mle_step(netF, optF, z_g_k, f_max_norm: Optional[float] = None, fused: bool = False):
    """train.py:404-415: one Adam step of the flow on the Langevin-inferred z.  Returns loss_f (detached).
    fused=False restates the reference line by line (autograd through `netF(...)`); fused=True computes the same
    loss and gradients with `netF.mle_grads` (no autograd graph, no element-wise torch launches)."""
    import numpy as np
    if fused:
        optF.zero_grad(set_to_none=True)
        loss_f = netF.mle_grads(z_g_k.reshape(z_g_k.shape[0], -1), max_norm=f_max_norm,    # clip: train.py:413-414
                                reuse_buffers=True)
        optF.step()
        return loss_f
    optF.zero_grad()
    z2d = torch.squeeze(z_g_k)
    z1, logdet, _ = netF(z2d, objective=torch.zeros(int(z_g_k.shape[0]), device=z2d.device), init=False)
    prior_ll = -0.5 * (z1 ** 2)
    prior_ll = prior_ll.flatten(1).sum(-1) + np.log(2 * np.pi)
    ll = prior_ll + logdet
    loss_f = -ll.mean()
    loss_f.backward()
    if f_max_norm is not None:
        torch.nn.utils.clip_grad_norm_(netF.parameters(), f_max_norm)                        # train.py:413-414
    optF.step()
    return loss_f.detach()


class GraphedLangevinSampler:
    """The K-step sampler above with ONE step captured in a HIP graph and replayed K times (train.py:311-329).

    At the reference's batch size (B = 100) a Langevin step is ~20 small launches (generator forward + input
    gradient through MIOpen, the flow's two kernels) and the host cannot issue them as fast as the GPU retires
    them; a graph replay issues the whole step with one call.  The step works on static buffers (z is updated in
    place by the fused flow update), the Langevin noise is drawn inside the update kernel from a device-side
    counter that the graph itself advances (`PhiloxNoise(offset_dev=)`), and the flow's prepared weights live in a
    buffer that `netF._plan()` refreshes in place -- so optimizer steps between `run()` calls need no re-capture.

        sampler = GraphedLangevinSampler(netG, netF, B, nz, x_shape, g_l_step_size=0.1, g_llhd_sigma=0.3, seed=1)
        z_k, gg_norm, gf_norm, f = sampler.run(z0, x, g_l_steps=20, offset=it * 20)
    """

    def __init__(self, netG: nn.Module, netF, B: int, nz: int, x_shape, *, g_l_step_size: float, g_llhd_sigma: float,
                 g_l_with_noise: bool = True, seed: int = 0, row0: int = 0, device=None, warmup: int = 3):
        from . import flow
        dev = device or next(netF.parameters()).device
        self.netG, self.netF, self.B, self.nz = netG, netF, B, nz
        self.z = torch.zeros(B, nz, 1, 1, device=dev)
        self.x = torch.zeros(tuple(x_shape), device=dev)
        self.ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.noise = flow.PhiloxNoise(seed, 0, row0, offset_dev=self.ctr) if g_l_with_noise else None
        self.s, self.sigma = float(g_l_step_size), float(g_llhd_sigma)
        self.mse = nn.MSELoss(reduction="sum")
        netF._plan()                                   # prepared weights exist before anything is captured
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                  # warm-up off the capture stream (MIOpen finds its solvers here)
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.f_log_lkhd, self.gg_norm, self.gf_norm = self._step()

    def _step(self):
        zr = self.z.detach().requires_grad_(True)
        g_log_lkhd = 1.0 / (2.0 * self.sigma * self.sigma) * self.mse(self.netG(zr), self.x)         # train.py:312-313
        z_grad_g = torch.autograd.grad(g_log_lkhd, zr)[0]                                            # train.py:314
        _, ll, gf, gg = self.netF.langevin_step(self.z.view(self.B, self.nz), z_grad_g.reshape(self.B, self.nz),
                                                self.noise, self.s, inplace=True)                    # :316-326, in place
        self.ctr.add_(1)                                                                             # next step's noise
        return -ll.sum(), gg.mean(), gf.mean()

    def run(self, z0: torch.Tensor, x: torch.Tensor, g_l_steps: int, offset: int = 0):
        """Returns (z_k (B,nz,1,1), mean |z_grad_g|, mean |z_grad_f|, f_log_lkhd of the last step's input)."""
        self.z.copy_(z0.reshape(self.z.shape))
        self.x.copy_(x)
        self.ctr.fill_(int(offset))
        self.netF._plan()                              # re-derives the prepared weights in place if a parameter changed
        for _ in range(g_l_steps):
            self.graph.replay()
        return self.z.detach().clone(), self.gg_norm, self.gf_norm, self.f_log_lkhd
