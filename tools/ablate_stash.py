"""GPU: time of the forward with stash at B = 65 536 for the library named by LSNF_LIB_PATH (tools/ablate_stash.sh)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
F.set_small_batch_max(0)
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
zd = torch.randn(B, bench.NZ, device=dev)
act = F.new_act_saved(plan, B, dev)
outs = (torch.empty_like(zd), torch.empty(B, device=dev), torch.empty(B, device=dev))
saved = torch.empty(bench.DEPTH - 1, B, bench.NZ, device=dev)
def t_us(fn, n=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[3]
a = t_us(lambda: lsnf_amd.forward(plan, zd, out=outs))
b = t_us(lambda: lsnf_amd.forward(plan, zd, out=outs, act_saved=act, z_saved_out=saved))
gg = torch.randn(B, bench.NZ, device=dev); nn_ = torch.randn(B, bench.NZ, device=dev)
c = t_us(lambda: F.langevin_step(plan, zd, gg, nn_, 0.1, reuse_buffers=True))
print(f"{os.environ.get('LSNF_LIB_PATH', 'default')}: B={B} forward {a:6.1f} us  forward+stash {b:6.1f} us  Langevin step {c:6.1f} us", flush=True)
