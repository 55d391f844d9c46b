"""GPU: randomised consistency sweep of the parameter-gradient FAST path at large batches (forward with stash + h dump, backward from
the stash with the g dump, batch contraction of lsnf_params3.hip on the bf16 pipe -- row-major or tiled dump, every segment shape)
against the recomputing fp32 path of the same library on the same inputs.  A layout bug shows as an O(1) difference; ReLU-kink flips
(one sample of tens of thousands) as <= ~1e-3 of a tensor's norm.   python tools/fuzz_params_fast.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow as F
from oracle import flow_oracle as O
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
fails, checks, t0 = 0, 0, time.time()
for case in range(n_cases):
    nz = int(rng.choice([8, 16, 24, 40, 48, 64, 72, 96, 104, 120, 128]))
    width = int(rng.choice([4, 12, 16, 20, 32, 48, 64, 96, 128]))
    depth = int(rng.integers(1, 4))
    B = int(rng.choice([12288, 13001, 16384, 16385, 20000, 33000, 40001]))
    small = int(rng.choice([F.SMALL_BATCH_AUTO, 0]))
    prev = F.set_small_batch_max(small)
    try:
        p = O.init_params(nz, width, depth, seed=100 + case)
        params = lsnf_amd.params_from_state_dict(p, depth, dev)
        plan = lsnf_amd.prepare(params, nz, width, depth)
        z = torch.randn(B, nz, generator=torch.Generator().manual_seed(case)).to(dev)
        act = F.new_act_saved(plan, B, dev); act.fill_(float("nan"))
        ws = F.new_params_workspace(plan, B, dev); ws.fill_(float("nan"))
        z1, _, _, saved = lsnf_amd.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
        fast = [g.clone() for g in lsnf_amd.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, act_saved=act, workspace=ws)]
        slow = lsnf_amd.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B)
        worst, where = 0.0, None
        for k, (a, b) in enumerate(zip(fast, slow)):
            checks += 1
            e = (a - b).norm().item() / max(b.norm().item(), 1e-12) if torch.isfinite(a).all() else float("inf")
            if e > worst: worst, where = e, F.BLOCK_PARAM_KEYS[k % 12] + f"[block {k // 12}]"
        bad = not (worst <= 2e-3)
        fails += bad
        print(("FAIL " if bad else "ok   ") + f"case{case} nz={nz} w={width} d={depth} B={B} small_max={small}: worst rel {worst:.2e} at {where}", flush=True)
    finally:
        F.set_small_batch_max(prev)
print(f"SUMMARY: {n_cases} cases, {checks} tensors compared, {fails} failing cases, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
