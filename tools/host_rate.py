"""How fast can the host enqueue bench.py's step?  (step = forward launch pair + stats bookkeeping on alternating streams)
Prints host us/step for the enqueue loop alone (no sync inside) next to the GPU-bound us/step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, lsnf_amd
from lsnf_amd import parallel
dev = torch.device("cuda:0")
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(bench.B_PER_GPU, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
zs = torch.randn(64, bench.NZ, device=dev)   # tiny batch through the same python path: GPU time ~0, host time the same
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def mk(zz):
    return [(torch.empty_like(zz), torch.empty(zz.shape[0], device=dev), torch.empty(zz.shape[0], device=dev)) for _ in range(2)]
red = [parallel.PipelinedStatsReducer(dev, bucket=32) for _ in range(2)]
def run(zz, outs, n):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            st = red[k].next_buffer()
            lsnf_amd.forward(plan, zz, out=outs[k], stats=st)
            red[k].submit(st)
for zz, tag in ((z, "B=65536"), (zs, "B=64 (latency kernel: host path only)")):
    outs = mk(zz)
    run(zz, outs, 600); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(zz, outs, 2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: enqueue {1e6*(t1-t0)/2000:.1f} us/step, total {1e6*(t2-t0)/2000:.1f} us/step")
