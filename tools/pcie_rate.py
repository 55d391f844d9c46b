"""PCIe-inclusive rate of the headline launch for a caller that hands over HOST buffers (not the library's boundary:
DESIGN.md section 5): pinned z -> device, forward, z1 / logdet / ll -> pinned host, serial and double-buffered."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
B, nz = bench.B_PER_GPU, bench.NZ
zh = torch.randn(B, nz).pin_memory()
oh = [(torch.empty(B, nz).pin_memory(), torch.empty(B).pin_memory(), torch.empty(B).pin_memory()) for _ in range(2)]
zd = [torch.empty(B, nz, device=dev) for _ in range(2)]
od = [(torch.empty(B, nz, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def one(k):
    with torch.cuda.stream(streams[k]):
        zd[k].copy_(zh, non_blocking=True)
        lsnf_amd.forward(plan, zd[k], out=od[k])
        for h, d in zip(oh[k], od[k]): h.copy_(d, non_blocking=True)
for nstr, tag in ((1, "serial (one stream)"), (2, "double-buffered (two streams)")):
    for i in range(20): one(i % nstr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for i in range(n): one(i % nstr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag}: {dt*1e3:.3f} ms per 65536-row batch -> {B/dt:.3e} latent-samples/s PCIe-inclusive")
