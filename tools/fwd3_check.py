"""Split-bf16 forward (LSNF_MATH_BF16X3) vs the fp32-MFMA forward and the float64 oracle, plus timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
from oracle import flow_oracle as O
dev = torch.device("cuda:0")
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
z = torch.randn(B, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
def timeit(n=300, warm=700):
    for _ in range(warm): lsnf_amd.forward(plan, z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lsnf_amd.forward(plan, z)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lsnf_amd.flow.set_math_mode(0)
z1a, lda, lla, _ = lsnf_amd.forward(plan, z)
ta = timeit()
lsnf_amd.flow.set_math_mode(1)
z1b, ldb, llb, _ = lsnf_amd.forward(plan, z)
tb = timeit()
print(f"fp32 MFMA {ta:.1f} us   bf16x3 {tb:.1f} us   speedup {ta / tb:.2f}x")
print("max |z1 diff|", (z1a - z1b).abs().max().item(), " max rel ll diff", ((lla - llb).abs() / lla.abs().clamp_min(1)).max().item())
# float64 truth on a sample of rows
idx = torch.arange(0, B, max(1, B // 2048))
pd = {}
for i in range(bench.DEPTH):
    for j, k in enumerate(lsnf_amd.flow.BLOCK_PARAM_KEYS):
        t = w[i * 12 + j]
        pd[O.block_prefix(i) + k] = t.reshape(1, -1) if t.dim() == 1 else t
p64 = O.to_dtype(pd, torch.float64)
if True:
    _, _, ll64 = O.flow_log_prob(p64, z[idx].cpu().double())
    for nm, ll in (("fp32 MFMA", lla), ("bf16x3", llb)):
        rel = ((ll[idx.to(dev)].cpu().double() - ll64).abs() / ll64.abs().clamp_min(1.0))
        print(f"{nm}: ll vs float64 oracle: max rel {rel.max().item():.3e} median {rel.median().item():.3e}")
