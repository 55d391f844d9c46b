import os, sys, torch
sys.path.insert(0, "/root/repo")
import lsnf_amd as lsnf
from oracle import flow_oracle as O
dev = torch.device("cuda:0")
nz, width, depth, B = 100, 128, 5, 4200
p = O.init_params(nz, width, depth, seed=3)
params = lsnf.params_from_state_dict(p, depth, dev)
plan = lsnf.prepare(params, nz, width, depth)
z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B)).to(dev)
ref = O.grad_neg_mean_ll_wrt_params(O.to_dtype(p, torch.float64), z.cpu().double())
keys = [O.block_prefix(i) + k for i in range(depth) for k in lsnf.flow.BLOCK_PARAM_KEYS]
z1, _, _, saved = lsnf.forward(plan, z, want_ll=False, save_for_backward=True)
for env in ("", "1"):
    if env: os.environ["LSNF_TN_PLAIN"] = "1"
    slow = lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B)
    worst = sorted(((float((g.cpu().double() - ref[k].reshape(g.shape)).norm() / max(ref[k].norm().item(), 1e-9)), k) for k, g in zip(keys, slow)), reverse=True)[:4]
    print("plain" if env else "lds", worst)
