import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import bench, lsnf_amd
dev = torch.device("cuda:0")
for (nz, w) in ((128, 64), (100, 64), (100, 128)):
    import numpy as np, types
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    torch.manual_seed(1); np.random.seed(1)
    net = lsnf_amd._netF(hps, nz=nz).to(dev)
    params = [p.detach() for p in net._param_list()]
    for _ in range(30):
        plan = lsnf_amd.prepare(params, nz, w, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        plan = lsnf_amd.prepare(params, nz, w, 5)
    e1.record(); torch.cuda.synchronize()
    print(f"nz={nz} w={w}: prepare {e0.elapsed_time(e1)/50*1e3:.1f} us per call", flush=True)
