#!/usr/bin/env python3
"""GPU: A/B of builds of liblsnf_flow.so on the NON-headline throughput kernels at B = 65 536 (forward with stash, backward from
the stash, reverse; C3 nz=128 w=64 and C5 nz=100 w=128), alternating child processes (LSNF_LIB_PATH), each timing = median of
9 chunks of 40 calls after a 300-launch clock ramp, buffers preallocated.   usage: ab_secondary.py lib1.so lib2.so ... [rounds]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, types, torch, numpy as np
sys.path.insert(0, os.getcwd())
import lsnf_amd
from lsnf_amd import flow
dev = torch.device("cuda:0")
flow.set_small_batch_max(0)
def med(fn, chunks=9, per=40):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) / per * 1e3)
    return sorted(r)[len(r) // 2]
out = []
for tag, nz, w in (("C3", 128, 64), ("C5", 100, 128)):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    torch.manual_seed(1); np.random.seed(1)
    net = lsnf_amd._netF(hps, nz=nz)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if ".fc_zeros." in n_: p_.add_(0.05 * torch.randn_like(p_))
    net = net.to(dev); plan = net._plan()
    B = 65536
    z = torch.randn(B, nz, device=dev)
    act = flow.new_act_saved(plan, B, dev)
    outs = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
    zs = torch.empty(plan.depth - 1, B, nz, device=dev)
    f = lambda: flow.forward(plan, z, save_for_backward=True, act_saved=act, out=outs, z_saved_out=zs)
    z1, ld, ll, saved = f()
    t_f = med(f)
    t_b = med(lambda: flow.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act))
    t_r = med(lambda: flow.reverse(plan, z))
    t_p = med(lambda: flow.forward(plan, z, out=outs))
    out.append(f"{tag}: fwd+stash {t_f:6.1f}  bwd(stash) {t_b:6.1f}  reverse {t_r:6.1f}  fwd {t_p:6.1f}")
print(" | ".join(out), flush=True)
'''


def main():
    args = sys.argv[1:]
    rounds = 2
    if args and args[-1].isdigit():
        rounds = int(args.pop())
    for r in range(rounds):
        for lib in args:
            env = dict(os.environ, LSNF_LIB_PATH=os.path.abspath(lib))
            o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=400)
            line = [l for l in o.stdout.splitlines() if l.startswith("C3")]
            print(f"{os.path.basename(lib):20s} {line[0] if line else 'FAILED: ' + o.stderr[-600:]}", flush=True)


main()
