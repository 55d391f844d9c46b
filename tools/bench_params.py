"""GPU: the parameter-gradient path (train.py:404-411) -- from-the-stash fast path vs the recomputing fp32 path, and the
recomputing backward w.r.t. z, at the reference's batch size and at the headline size.  Times are HIP-event medians."""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow
dev = torch.device("cuda:0")


def timeit(fn, n=200, warm=30, chunks=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    per = max(1, n // chunks); res = []
    for _ in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / per * 1e3)
    return sorted(res)[len(res) // 2]


def make(nz, w):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    torch.manual_seed(1); np.random.seed(1)
    net = lsnf_amd._netF(hps, nz=nz)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if ".fc_zeros." in n_: p_.add_(0.05 * torch.randn_like(p_))
    return net.to(dev)


out = []
for tag, nz, w, B in (("nz=100 w=64 B=100", 100, 64, 100), ("nz=128 w=64 B=100", 128, 64, 100), ("nz=100 w=128 B=100", 100, 128, 100),
                      ("nz=128 w=64 B=65536", 128, 64, 65536)):
    net = make(nz, w); plan = net._plan(); params = [p.detach() for p in net._param_list()]
    z = torch.randn(B, nz, device=dev)
    n = 300 if B <= 1000 else 30
    act = flow.new_act_saved(plan, B, dev); ws = flow.new_params_workspace(plan, B, dev)
    r = {"config": tag}
    z1, _, _, saved = flow.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
    r["forward_plain_us"] = timeit(lambda: flow.forward(plan, z, want_ll=False, save_for_backward=True), n)
    r["forward_stash_hdump_us"] = timeit(lambda: flow.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws), n)
    r["backward_params_fast_us"] = timeit(lambda: flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True, act_saved=act, workspace=ws), n)
    plan.__dict__.pop("_bp_state", None)
    r["backward_params_recompute_fp32_us"] = timeit(lambda: flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True), n)
    r["backward_z_from_stash_us"] = timeit(lambda: flow.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act), n)
    r["backward_z_recompute_us"] = timeit(lambda: flow.backward_z(plan, z1, saved, ll_scale=-1.0), n)
    r["mle_grads_fused_reuse_us"] = timeit(lambda: net.mle_grads(z, reuse_buffers=True), max(20, n // 3), 5)
    out.append(r); print(json.dumps(r), flush=True)
json.dump(out, open("gpurun_out/bench_params.json", "w"), indent=1)
