import os, sys, types
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=128).to(dev)
for B in (100, 65536):
    plan = net._plan(); params = [p.detach() for p in net._param_list()]
    z = torch.randn(B, 128, device=dev)
    act = flow.new_act_saved(plan, B, dev); ws = flow.new_params_workspace(plan, B, dev)
    for _ in range(30):
        z1, _, _, saved = flow.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
        flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True, act_saved=act, workspace=ws)
    torch.cuda.synchronize()
