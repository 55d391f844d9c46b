import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
def t_us(z, st, n=400):
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    for _ in range(600): lsnf_amd.forward(plan, z, out=outs, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lsnf_amd.forward(plan, z, out=outs, stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (65536, 32768, 16384):
    z = torch.randn(B, bench.NZ, device=dev)
    st = F.new_stats(dev)
    r = [t_us(z, None), t_us(z, st), t_us(z, None), t_us(z, st)]
    print(f"B={B}: no stats {r[0]:.1f} {r[2]:.1f}   with stats {r[1]:.1f} {r[3]:.1f}", flush=True)
