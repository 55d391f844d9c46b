"""GPU: kernel-family crossover of the other entry points (backward from the stash, reverse, Langevin step, forward with stash)
at 4 096 ... 32 768 rows, default arithmetic: latency family forced vs throughput family forced."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
def t_us(fn, n=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (4096, 8192, 12288, 16384, 24576, 32768):
    z = torch.randn(B, bench.NZ, device=dev)
    row = [f"B={B:6d}"]
    for fam, smax in (("latency", 1 << 30), ("throughput", 0)):
        F.set_small_batch_max(smax)
        act = F.new_act_saved(plan, B, dev)
        outs = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
        zs = torch.empty(plan.depth - 1, B, bench.NZ, device=dev)
        f = lambda: F.forward(plan, z, save_for_backward=True, act_saved=act, out=outs, z_saved_out=zs)
        z1, ld, ll, saved = f()
        tf = t_us(f)
        tb = t_us(lambda: F.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act))
        tr = t_us(lambda: F.reverse(plan, z))
        row.append(f"{fam}: fwd+stash {tf:6.1f} bwd {tb:6.1f} rev {tr:6.1f}")
    print("   ".join(row), flush=True)
