"""Throughput of back-to-back forward launches on one stream vs alternating over two streams (independent batches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, lsnf_amd
dev = torch.device("cuda:0")
w = [t.to(dev) for t in bench.synth_weights(1)]
plan = lsnf_amd.prepare(w, 128, 64, 5)
B = 65536
zs = [torch.randn(B, 128, device=dev) for _ in range(2)]
outs = [(torch.empty_like(zs[0]), torch.empty(B, device=dev), torch.empty(B, device=dev)) for _ in range(2)]
stats = [lsnf_amd.flow.new_stats(dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, nstreams):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            lsnf_amd.forward(plan, zs[k], out=outs[k], stats=stats[k])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for ns in (1, 2, 1, 2):
    run(500, ns)
    print(f"{ns} stream(s): {run(3000, ns):.1f} us per launch")
