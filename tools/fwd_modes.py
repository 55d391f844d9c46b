"""Throughput forward at the headline size in every arithmetic mode: kernel time (HIP events) and log-prob error vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd
from lsnf_amd import flow
from oracle import flow_oracle as O
dev = torch.device("cuda:0")
nz, w, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 64, 65536)
p = O.init_params(nz, w, 5, seed=3)
plan = flow.prepare(flow.params_from_state_dict(p, 5, dev), nz, w, 5)
z = torch.randn(B, nz, generator=torch.Generator().manual_seed(1))
_, _, ll64 = O.flow_log_prob(O.to_dtype(p, torch.float64), z[:8192].double())
zd = z.to(dev)
for _ in range(300): flow.forward(plan, zd)
for name, mode in (("fp32", flow.MATH_FP32), ("bf16x3", flow.MATH_BF16X3), ("bf16x3_32", flow._MATH_X_BF16X3_32), ("fp16x2", flow.MATH_FP16X2)):
    flow.set_math_mode(mode)
    for _ in range(200): out = flow.forward(plan, zd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(200): out = flow.forward(plan, zd)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200 * 1e3)
    ll = out[2][:8192].cpu().double()
    err = ((ll - ll64).abs() / ll64.abs().clamp_min(1.0))
    print(f"{name:10s} {np.median(ts):8.1f} us/launch (incl. launch gaps)   ll err vs f64: max {err.max():.3e} median {err.median():.3e}", flush=True)
