"""GPU: the 16 / 32 / 64-rows-per-workgroup forms of the latency forward (lsnf_small3_fwd_kernel<C, ST>): bit-level agreement
of the three shapes with each other (same arithmetic order per row), sampled rows against the oracle, the stash they write
against the throughput kernel's, and their time at the shard sizes.  Run once per shape: LSNF_SMALL3_ST=1|2|4 (unset: default dispatch)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
from oracle import flow_oracle as O
F = lsnf_amd.flow
dev = torch.device("cuda:0")
p = bench.synth_state_dict(1) if hasattr(bench, "synth_state_dict") else None
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
print("LSNF_SMALL3_ST =", os.environ.get("LSNF_SMALL3_ST"))
def t_us(B, stats, n=300):
    z = torch.randn(B, bench.NZ, device=dev)
    outs = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
    st = F.new_stats(dev) if stats else None
    for _ in range(300): lsnf_amd.forward(plan, z, out=outs, stats=st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): lsnf_amd.forward(plan, z, out=outs, stats=st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
torch.manual_seed(0)
for B in (100, 777, 4096, 5000, 8192, 9001, 16384):
    z = torch.randn(B, bench.NZ, device=dev)
    F.set_small_batch_max(1 << 30)
    z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True)
    F.set_small_batch_max(0)
    z1t, ldt, llt, savedt = lsnf_amd.forward(plan, z, save_for_backward=True)
    torch.cuda.synchronize()
    rel = ((ll - llt).abs() / llt.abs()).max().item()
    dz = (z1 - z1t).abs().max().item()
    ds = (saved - savedt).abs().max().item() if saved is not None and savedt is not None else float("nan")
    print(f"B={B:6d}: latency vs throughput kernel: ll rel {rel:.2e}  z1 abs {dz:.2e}  z_saved abs {ds:.2e}", flush=True)
F.set_small_batch_max(1 << 30)
for B in (100, 4096, 6144, 8192, 12288, 16384):
    print(f"B={B:6d}: latency forward {t_us(B, False):6.1f} us, with in-kernel sums {t_us(B, True):6.1f} us", flush=True)
