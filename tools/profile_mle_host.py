"""Host-side profile (cProfile) of netF.mle_grads at the reference's batch size: where the wall time beyond the ~150 us
of GPU work goes."""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=100).to(dev)
z = torch.randn(100, 100, device=dev)
import time
def wall(fn, n=400):
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def fresh():
    net.zero_grad(set_to_none=True); net.mle_grads(z)
def reused():
    net.mle_grads(z, reuse_buffers=True)
print("wall us/call: fresh buffers %.1f, reused buffers %.1f" % (wall(fresh), wall(reused)))
for name, fn in (("fresh", fresh), ("reused", reused)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    print("====", name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
