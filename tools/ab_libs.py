#!/usr/bin/env python3
"""GPU: A/B of two builds of liblsnf_flow.so in one job (alternating child processes, LSNF_LIB_PATH): kernel-only time of the
headline forward (and optionally other entry points) per math mode.   usage: ab_libs.py libA.so libB.so [rounds]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
F.set_small_batch_max(0)
z = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
def t_us(fn, n=400):
    for _ in range(600): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
for name, mode in (("fwd3b", getattr(F, "MATH_BF16X3_PHASED", F.MATH_BF16X3)), ("fwd3p", F._MATH_X_BF16X3_PIPE), ("fwd3q", F.MATH_BF16X3)):
    F.set_math_mode(mode)
    res.append(f"{name} {t_us(lambda: lsnf_amd.forward(plan, z, out=outs)):.2f}")
F.set_math_mode(F.MATH_BF16X3)
z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True)[:4] if False else (None, None, None, None)
print("  ".join(res), flush=True)
'''


def main():
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, LSNF_LIB_PATH=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
            line = [l for l in out.stdout.splitlines() if l.startswith("fwd3b")]
            print(f"{os.path.basename(lib):24s} {line[0] if line else 'FAILED: ' + out.stderr[-400:]}", flush=True)


main()
