"""Profiling driver: N launches of the headline forward (nz=128, w=64, depth=5, B=65536)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, lsnf_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(bench.B_PER_GPU, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
out = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
for _ in range(3):
    lsnf_amd.forward(plan, z, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    lsnf_amd.forward(plan, z, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"lib={os.environ.get('LSNF_LIB_PATH','default')} fwd {ms*1e3:.1f} us/launch -> {bench.FLOP_PER_SAMPLE*bench.B_PER_GPU/ms/1e9:.1f} TFLOP/s, ll[0]={out[2][0].item():.6f}")
