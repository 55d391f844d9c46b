import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
def t_us(z, n=300):
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    st = F.new_stats(dev)
    for _ in range(400): lsnf_amd.forward(plan, z, out=outs, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lsnf_amd.forward(plan, z, out=outs, stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (65536, 32768, 16384, 8192, 4096):
    z = torch.randn(B, bench.NZ, device=dev)
    F.set_small_batch_max(F.SMALL_BATCH_AUTO); a = t_us(z)
    F.set_small_batch_max(0); b = t_us(z)
    F.set_small_batch_max(1 << 30); c = t_us(z)
    print(f"B={B:6d} (= 65536/{65536//B})  default {a:6.1f} us   throughput kernel {b:6.1f}   latency kernel {c:6.1f}   -> rows/s per GPU {B/min(a,b,c)*1e6:.3e}  x{65536//B} = {B/min(a,b,c)*1e6*(65536//B):.3e}", flush=True)
