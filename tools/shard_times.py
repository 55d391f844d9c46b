"""GPU: forward time of one GPU on the shard sizes of strong scaling (65 536 rows over N GPUs) and around the kernel-family
crossover, on the default dispatch, the throughput kernel forced and the latency kernel forced.  `nostats`: without in-kernel sums."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
USE_STATS = "nostats" not in sys.argv
def t_us(z, n=300):
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    st = F.new_stats(dev) if USE_STATS else None
    for _ in range(400): lsnf_amd.forward(plan, z, out=outs, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lsnf_amd.forward(plan, z, out=outs, stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("forward + log-prob,", "with in-kernel sums" if USE_STATS else "without in-kernel sums")
for B in (65536, 32768, 16384, 12288, 10240, 8192, 6144, 4096):
    z = torch.randn(B, bench.NZ, device=dev)
    F.set_small_batch_max(F.SMALL_BATCH_AUTO); a = t_us(z)
    F.set_small_batch_max(0); b = t_us(z)
    F.set_small_batch_max(1 << 30); c = t_us(z)
    print(f"B={B:6d} (65536/{65536/B:g})  default {a:6.1f} us   throughput kernel {b:6.1f}   latency kernel {c:6.1f}   -> rows/s per GPU {B/min(a,b,c)*1e6:.3e}  x{65536/B:g} = {B/min(a,b,c)*1e6*(65536/B):.3e}", flush=True)
