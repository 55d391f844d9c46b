"""Default dispatch of lsnf_forward around the thresholds (12 288 rows in LSNF_MATH_FP16X2 while the threshold is the
built-in default, else 16 384): time per launch and log-prob error vs the float64 oracle on both sides."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lsnf_amd
from lsnf_amd import flow
from oracle import flow_oracle as O
dev = torch.device('cuda:0')
p = O.init_params(128, 64, 5, seed=4)
plan = flow.prepare(flow.params_from_state_dict(p, 5, dev), 128, 64, 5)
for B in (12288, 12289, 14000, 16384, 16385):
    z = torch.randn(B, 128, generator=torch.Generator().manual_seed(B))
    zd = z.to(dev)
    for _ in range(20): out = flow.forward(plan, zd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): out = flow.forward(plan, zd)
    e1.record(); torch.cuda.synchronize()
    idx = torch.arange(0, B, 53)
    _, _, llr = O.flow_log_prob(O.to_dtype(p, torch.float64), z[idx].double())
    err = ((out[2].cpu()[idx].double() - llr).abs() / llr.abs()).max().item()
    print(B, "%.1f us" % (e0.elapsed_time(e1) / 200 * 1e3), "ll err %.2e" % err)
