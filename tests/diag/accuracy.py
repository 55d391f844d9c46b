"""Error statistics of the HIP forward/backward vs float64 oracle, at the headline geometry (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lsnf_amd
from oracle import flow_oracle as O
dev = torch.device("cuda:0")
for (nz, w, seed, std, sig) in [(128, 64, 1, 0.0, 1.0), (128, 64, 2, 0.05, 3.0), (100, 128, 3, 0.0, 1.0)]:
    p = O.init_params(nz, w, 5, seed=seed, fcz_std=0.05 if std == 0 else 0.1, all_std=std)
    z = sig * torch.randn(2048, nz, generator=torch.Generator().manual_seed(7))
    p64 = O.to_dtype(p, torch.float64)
    z1r, ldr, llr = O.flow_log_prob(p64, z.double())
    z1c, ldc, llc = O.flow_log_prob(p, z)
    plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(p, 5, dev), nz, w, 5)
    z1, ld, ll, saved = lsnf_amd.forward(plan, z.to(dev), save_for_backward=True)
    rel = lambda a, b: ((a.double() - b).abs() / b.abs()).max().item()
    print(f"nz={nz} w={w} sig={sig} all_std={std}: ll rel err vs f64: HIP {rel(ll.cpu(), llr):.2e}  torch-CPU-f32 {rel(llc, llr):.2e} | "
          f"z1 abs: HIP {(z1.cpu().double()-z1r).abs().max().item():.2e} CPU {(z1c.double()-z1r).abs().max().item():.2e}")
