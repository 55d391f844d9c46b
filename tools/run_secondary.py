"""Profiling driver for the NON-headline kernels: N back-to-back calls of each op at the reference's
batch size (B = 100) and at the headline size (B = 65 536), C3 geometry (nz=128, w=64, depth 5), default arithmetic
(bf16x3).  Run plain (prints HIP-event times as JSON), under `rocprofv3 --kernel-trace` (per-kernel trace averages) and
under `rocprofv3 --pmc` (instruction mix, HBM bytes): tools/round2_measure.sh; tools/secondary_roofline.py builds the table.

    python tools/run_secondary.py [N] [B]          # B: 100 or 65536 only (default: both)
"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only_B = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=128)
with torch.no_grad():
    for n_, p_ in net.named_parameters():
        if ".fc_zeros." in n_: p_.add_(0.05 * torch.randn_like(p_))
net = net.to(dev); plan = net._plan(); params = [p.detach() for p in net._param_list()]


def timed(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
for B in (100, 65536):
    if only_B and B != only_B:
        continue
    z = torch.randn(B, 128, device=dev); gg = torch.randn(B, 128, device=dev)
    act = flow.new_act_saved(plan, B, dev); ws = flow.new_params_workspace(plan, B, dev)
    outb = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
    saved = torch.empty((4, B, 128), device=dev)
    z1, _, _, _ = flow.forward(plan, z, out=outb, act_saved=act, z_saved_out=saved, params_ws=ws)
    n = n_calls if B > 1000 else 4 * n_calls
    r = {}
    r["forward"] = timed(lambda: flow.forward(plan, z, out=outb), n)
    r["forward_stash"] = timed(lambda: flow.forward(plan, z, out=outb, act_saved=act, z_saved_out=saved), n)
    r["forward_stash_hdump"] = timed(lambda: flow.forward(plan, z, out=outb, act_saved=act, z_saved_out=saved, params_ws=ws), n)
    r["reverse"] = timed(lambda: flow.reverse(plan, z), n)
    r["backward_z_from_stash"] = timed(lambda: flow.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act), n)
    r["backward_z_restash"] = timed(lambda: flow.backward_z(plan, z1, saved, ll_scale=-1.0), n)
    r["langevin_step"] = timed(lambda: net.langevin_step(z, gg, flow.PhiloxNoise(7, 3), 0.1, reuse_buffers=True), n)
    z1, _, _, _ = flow.forward(plan, z, out=outb, act_saved=act, z_saved_out=saved, params_ws=ws)
    r["backward_params_from_stash"] = timed(lambda: flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True, act_saved=act, workspace=ws), n)
    r["mle_grads"] = timed(lambda: net.mle_grads(z, reuse_buffers=True), max(10, n // 2))
    out[f"B={B}"] = r
print(json.dumps(out))
