#!/usr/bin/env python3
"""GPU: the 16x16x32 form of the software-pipelined bf16x3 forward (lsnf_fwd3q_kernel, the default of math mode BF16X3) against the phase-separated kernel (lsnf_fwd3b_kernel), the 32x32x16 pipeline and the fp32-MFMA kernel:
differences, then kernel-only times at the headline size (alternating, two rounds)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lsnf_amd

dev = torch.device("cuda:0")
F = lsnf_amd.flow
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
F.set_small_batch_max(0)
KERNELS = {"fwd3b (16x16x32, phases)": F.MATH_BF16X3_PHASED, "fwd3p (32x32x16, pipelined)": F._MATH_X_BF16X3_PIPE,
           "fwd3q (16x16x32, pipelined)": F.MATH_BF16X3, "fp32 MFMA": F.MATH_FP32}


def accepted(mode):          # (the 32x32x16 kernels exist in research builds only: make EXTRA=-DLSNF_EXPERIMENTAL_KERNELS)
    prev = F.set_math_mode(-1)
    F.set_math_mode(mode)
    ok = F.set_math_mode(-1) == mode
    F.set_math_mode(prev)
    return ok


KERNELS = {k: v for k, v in KERNELS.items() if accepted(v)}


def select(name):
    F.set_math_mode(KERNELS[name])


def run(z, name, stats=False):
    select(name)
    out = lsnf_amd.forward(plan, z)
    torch.cuda.synchronize()
    return out[0].clone(), out[1].clone(), out[2].clone()


def t_us(z, name, n=400):
    select(name)
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    for _ in range(600):
        lsnf_amd.forward(plan, z, out=outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lsnf_amd.forward(plan, z, out=outs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


d = lambda a, b: ((a - b).abs() / b.abs().clamp_min(1.0)).max().item()   # noqa: E731
for B in (1, 37, 300, 20000, 32768, 40000, 65536, 65537):
    z = torch.randn(B, bench.NZ, generator=torch.Generator().manual_seed(B)).to(dev)
    new = run(z, "fwd3q (16x16x32, pipelined)")
    old = run(z, "fwd3b (16x16x32, phases)")
    f32 = run(z, "fp32 MFMA")
    print(f"B={B:6d}  fwd3q vs fwd3b: z1 {d(new[0], old[0]):.2e} logdet {d(new[1], old[1]):.2e} ll {d(new[2], old[2]):.2e}   "
          f"fwd3q vs fp32: ll {d(new[2], f32[2]):.2e}   fwd3b vs fp32: ll {d(old[2], f32[2]):.2e}", flush=True)
z = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
for rnd in range(2):
    for name in ("fwd3b (16x16x32, phases)", "fwd3p (32x32x16, pipelined)", "fwd3q (16x16x32, pipelined)"):
        print(f"{name:32s} {t_us(z, name):8.2f} us per launch (B=65536)", flush=True)
