"""GPU: time of lsnf_backward_params at B = 65 536 (fast path) for the library named by LSNF_LIB_PATH (tools/ablate_x3.sh)."""
import os, sys, types
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=128).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
plan = net._plan(); params = [p.detach() for p in net._param_list()]
z = torch.randn(B, 128, device=dev)
act = flow.new_act_saved(plan, B, dev); ws = flow.new_params_workspace(plan, B, dev)
z1, _, _, saved = flow.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
fn = lambda: flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True, act_saved=act, workspace=ws)
for _ in range(20): fn()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"{os.environ.get('LSNF_LIB_PATH', 'default')} (LSNF_TN_X3={os.environ.get('LSNF_TN_X3')}): B={B} backward_params {sorted(ts)[3]:7.1f} us", flush=True)
