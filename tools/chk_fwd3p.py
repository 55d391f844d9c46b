#!/usr/bin/env python3
"""GPU: the software-pipelined bf16x3 forward (lsnf_fwd3p.hip) against the phase-separated one (lsnf_fwd3.hip, selected
with LSNF_NO_FWD3P=1) and the fp32-MFMA kernel: differences and kernel-only times at the headline size."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lsnf_amd

dev = torch.device("cuda:0")
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
lsnf_amd.flow.set_small_batch_max(0)


def run(z, mode, old):
    lsnf_amd.flow.set_math_mode(mode)
    if old:
        os.environ["LSNF_NO_FWD3P"] = "1"
    else:
        os.environ.pop("LSNF_NO_FWD3P", None)
    out = lsnf_amd.forward(plan, z)
    torch.cuda.synchronize()
    return out[0].clone(), out[1].clone(), out[2].clone()


def t_us(z, mode, old, n=300):
    lsnf_amd.flow.set_math_mode(mode)
    if old:
        os.environ["LSNF_NO_FWD3P"] = "1"
    else:
        os.environ.pop("LSNF_NO_FWD3P", None)
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    for _ in range(300):
        lsnf_amd.forward(plan, z, out=outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lsnf_amd.forward(plan, z, out=outs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (1, 37, 300, 20000, 32768, 40000, 65536, 65537):
    z = torch.randn(B, bench.NZ, generator=torch.Generator().manual_seed(B)).to(dev)
    new = run(z, lsnf_amd.flow.MATH_BF16X3, False)
    old = run(z, lsnf_amd.flow.MATH_BF16X3, True)
    f32 = run(z, lsnf_amd.flow.MATH_FP32, True)
    d = lambda a, b: ((a - b).abs() / b.abs().clamp_min(1.0)).max().item()   # noqa: E731
    print(f"B={B:6d}  new vs old: z1 {d(new[0], old[0]):.2e} logdet {d(new[1], old[1]):.2e} ll {d(new[2], old[2]):.2e}   "
          f"new vs fp32: ll {d(new[2], f32[2]):.2e}   old vs fp32: ll {d(old[2], f32[2]):.2e}", flush=True)
z = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
for name, mode, old in (("fwd3p (pipelined, 32x32x16)", lsnf_amd.flow.MATH_BF16X3, False), ("fwd3b (16x16x32)", lsnf_amd.flow.MATH_BF16X3, True),
                        ("fwd3 (32x32x16)", lsnf_amd.flow.MATH_BF16X3_32, True), ("fp32 MFMA", lsnf_amd.flow.MATH_FP32, True),
                        ("fwd3p again", lsnf_amd.flow.MATH_BF16X3, False)):
    print(f"{name:32s} {t_us(z, mode, old):8.2f} us per launch (B=65536)", flush=True)
