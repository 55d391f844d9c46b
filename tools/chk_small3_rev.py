"""GPU: latency reverse (lsnf_small3_rev_kernel<C, ST>): against the throughput reverse (lsnf_rev3.hip), the round trip through
the forward, sampled rows against the oracle, and time by batch size.  LSNF_SMALL3_ST=1|2|4 forces the workgroup shape."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
from oracle import flow_oracle as O
F = lsnf_amd.flow
dev = torch.device("cuda:0")
print("LSNF_SMALL3_ST =", os.environ.get("LSNF_SMALL3_ST"))
def t_us(fn, n=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for nz, width, depth in ((bench.NZ, bench.WIDTH, bench.DEPTH), (20, 32, 3), (100, 128, 5)):
    p = O.init_params(nz, width, depth, seed=3)
    plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(p, depth, dev), nz, width, depth)
    for B in (100, 777, 5000, 9001, 16384):
        z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B))
        zd = z.to(dev); ob = torch.randn(B, generator=torch.Generator().manual_seed(B + 1)).to(dev)
        F.set_small_batch_max(1 << 30)
        x, o = lsnf_amd.reverse(plan, zd, objective=ob)
        F.set_small_batch_max(0)
        xt, ot = lsnf_amd.reverse(plan, zd, objective=ob)
        F.set_small_batch_max(1 << 30)
        zb, ld, _, _ = lsnf_amd.forward(plan, x)
        torch.cuda.synchronize()
        idx = torch.arange(0, B, max(1, B // 32))
        xr, orf = O.flow_reverse(O.to_dtype(p, torch.float64), z[idx].double(), ob[idx].cpu().double())
        eo = ((x[idx].cpu() - xr).norm() / xr.norm()).item()
        print(f"nz={nz:3d} B={B:6d}: vs throughput reverse x abs {(x - xt).abs().max().item():.2e} obj abs {(o - ot).abs().max().item():.2e}  "
              f"round trip rel {((zb - zd).norm() / zd.norm()).item():.2e}  logdet sum abs {(o - ob + ld).abs().max().item():.2e}  oracle rel {eo:.2e}", flush=True)
p = O.init_params(bench.NZ, bench.WIDTH, bench.DEPTH, seed=3)
plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(p, bench.DEPTH, dev), bench.NZ, bench.WIDTH, bench.DEPTH)
for B in (100, 2048, 4096, 8192, 16384, 32768):
    zd = torch.randn(B, bench.NZ, device=dev)
    F.set_small_batch_max(1 << 30)
    a = t_us(lambda: lsnf_amd.reverse(plan, zd))
    F.set_small_batch_max(0)
    b = t_us(lambda: lsnf_amd.reverse(plan, zd))
    print(f"B={B:6d}: latency reverse {a:6.1f} us   throughput reverse {b:6.1f} us", flush=True)
