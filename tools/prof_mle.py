import os, sys, types, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lsnf_amd
from lsnf_amd import langevin
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
net = lsnf_amd._netF(hps, nz=100).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.999))
zk = torch.randn(100, 100, 1, 1, device=dev)
for _ in range(5): langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0)
torch.cuda.synchronize()
def timed(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("full mle step us", timed(lambda: langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0)))
z2d = zk.view(100, 100); obj = torch.zeros(100, device=dev)
def fb():
    opt.zero_grad()
    a, b, _ = net(z2d, objective=obj)
    (-(-0.5 * (a ** 2).sum(1) + 1.8378770664093453 + b).mean()).backward()
print("fwd+bwd only us", timed(fb))
print("opt.step only us", timed(lambda: opt.step()))
print("clip only us", timed(lambda: torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): fb()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
