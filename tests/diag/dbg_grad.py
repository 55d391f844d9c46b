import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from oracle import flow_oracle as O
import lsnf_amd
dev = torch.device('cuda:0')
p, g = load_golden("c3_nz128_w64_B200")
plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(p, 5, dev), 128, 64, 5)
z = torch.from_numpy(g["z"]).to(dev)
z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True)
gz = lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0).cpu().numpy()
ref = g["grad_z"]
d = np.abs(gz - ref)
print("max", d.max(), "rows with max err > 1e-5:", np.where(d.max(1) > 1e-5)[0], d.max(1)[d.max(1) > 1e-5])
p64 = O.to_dtype(p, torch.float64)
g64 = O.grad_neg_sum_ll_wrt_z(p64, torch.from_numpy(g["z"]).double()).numpy()
print("ref32 vs f64: relL2", np.linalg.norm(ref - g64)/np.linalg.norm(g64), "max", np.abs(ref-g64).max())
print("gpu   vs f64: relL2", np.linalg.norm(gz - g64)/np.linalg.norm(g64), "max", np.abs(gz-g64).max())
bad = np.where(np.abs(ref-g64).max(1) > 1e-5)[0]; print("ref rows bad vs f64", bad)
bad = np.where(np.abs(gz-g64).max(1) > 1e-5)[0]; print("gpu rows bad vs f64", bad)
