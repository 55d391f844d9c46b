"""GPU: latency backward / Langevin step (lsnf_small3_bwd_kernel<C, ST, DUMP>): against the oracle on sampled rows and timing.
LSNF_SMALL3_ST=1|2|4 forces the workgroup shape of forward and backward."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
from oracle import flow_oracle as O
F = lsnf_amd.flow
dev = torch.device("cuda:0")
nz, width, depth = bench.NZ, bench.WIDTH, bench.DEPTH
p = O.init_params(nz, width, depth, seed=3)
plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(p, depth, dev), nz, width, depth)
print("LSNF_SMALL3_ST =", os.environ.get("LSNF_SMALL3_ST"))
F.set_small_batch_max(1 << 30)
for B in (100, 777, 5000, 9001):
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B))
    zd = z.to(dev)
    act = F.new_act_saved(plan, B, dev)
    z1, ld, ll, saved = lsnf_amd.forward(plan, zd, save_for_backward=True, act_saved=act)
    gz = lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act).cpu()
    idx = torch.arange(0, B, max(1, B // 64))
    ref = O.grad_neg_sum_ll_wrt_z(p, z[idx])
    ok = O.relu_margin(p, z[idx]) > 2e-6
    err = ((gz[idx] - ref)[ok].norm() / ref[ok].norm()).item()
    print(f"B={B:6d}: grad_z rel-L2 vs oracle {err:.2e}", flush=True)
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
for B in (100, 4096, 8192, 16384):
    zd = torch.randn(B, nz, device=dev); gg = torch.randn(B, nz, device=dev); nn_ = torch.randn(B, nz, device=dev)
    act = F.new_act_saved(plan, B, dev)
    outs = (torch.empty_like(zd), torch.empty(B, device=dev), torch.empty(B, device=dev))
    saved = torch.empty(depth - 1, B, nz, device=dev)
    fw = lambda: lsnf_amd.forward(plan, zd, out=outs, act_saved=act, z_saved_out=saved)
    fw()
    bw = lambda: lsnf_amd.backward_z(plan, outs[0], saved, ll_scale=-1.0, act_saved=act)
    lv = lambda: F.langevin_step(plan, zd, gg, nn_, 0.1, reuse_buffers=True)
    print(f"B={B:6d}: forward+stash {t_us(fw):6.1f} us   backward from the stash {t_us(bw):6.1f} us   Langevin step {t_us(lv):6.1f} us", flush=True)
