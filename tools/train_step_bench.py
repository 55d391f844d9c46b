"""Full synthetic training iteration (BASELINE.json configs[1]: SVHN nz=100 ngf=64 g_l_steps=20, B=100) on one GPU:
K-step Langevin sampling (generator gradient by torch autograd + fused flow step), generator Adam step, flow-MLE
Adam step -- the loop of reference train.py:376-415 with synthetic x.  Reports ms/iteration and the flow's share.
The generator is a stock PyTorch ConvTranspose2d stack of the SVHN shape (out of scope of the HIP build)."""
import json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
import lsnf_amd
from lsnf_amd import langevin

dev = torch.device("cuda:0")
NZ, NGF, NC, B, K, S, SIGMA = 100, 64, 3, 100, 20, 0.1, 0.3

def svhn_generator():
    f = lambda: nn.LeakyReLU(0.2)
    return nn.Sequential(nn.ConvTranspose2d(NZ, NGF * 8, 4, 1, 0), f(), nn.ConvTranspose2d(NGF * 8, NGF * 4, 4, 2, 1), f(),
                         nn.ConvTranspose2d(NGF * 4, NGF * 2, 4, 2, 1), f(), nn.ConvTranspose2d(NGF * 2, NC, 4, 2, 1), nn.Tanh())

hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
netG = svhn_generator().to(dev)
netF = lsnf_amd._netF(hps, nz=NZ).to(dev)
optG = torch.optim.Adam(netG.parameters(), lr=3e-4, betas=(0.5, 0.999))
optF = torch.optim.Adam(netF.parameters(), lr=1e-4, betas=(0.5, 0.999))
mse = nn.MSELoss(reduction="sum")
x = torch.tanh(torch.randn(B, NC, 32, 32, device=dev))

def iteration():
    z0 = torch.randn(B, NZ, 1, 1, device=dev)
    zk, ggn, gfn, f = langevin.sample_langevin_post_z_with_flow(z0, x, netG, netF, g_l_steps=K, g_l_step_size=S,
                                                                g_llhd_sigma=SIGMA, g_l_with_noise=True)
    optG.zero_grad()
    loss_g = mse(netG(zk), x) / B                      # train.py:391-393
    loss_g.backward(); optG.step()
    loss_f = langevin.flow_mle_step(netF, optF, zk, f_max_norm=100.0)
    return loss_g, loss_f

def sync_time(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): iteration()
ms_iter = sync_time(iteration, 10)
z2d = torch.randn(B, NZ, device=dev); gg = torch.randn(B, NZ, device=dev); nn_ = torch.randn(B, NZ, device=dev)
ms_flow_step = sync_time(lambda: netF.langevin_step(z2d, gg, nn_, S), 200)
zk = torch.randn(B, NZ, 1, 1, device=dev)
ms_mle = sync_time(lambda: langevin.flow_mle_step(netF, optF, zk, f_max_norm=100.0), 20)
def gstep():
    z = torch.randn(B, NZ, 1, 1, device=dev, requires_grad=True)
    g = 1.0 / (2 * SIGMA * SIGMA) * mse(netG(z), x)
    torch.autograd.grad(g, z)
ms_g = sync_time(gstep, 100)
res = {"config": f"SVHN nz={NZ} ngf={NGF} g_l_steps={K} B={B} (BASELINE.json configs[1]), synthetic x",
       "ms_per_iteration": ms_iter, "iterations_per_s": 1e3 / ms_iter,
       "flow_langevin_step_ms": ms_flow_step, "generator_grad_step_ms": ms_g, "flow_mle_step_ms": ms_mle,
       "flow_share_of_iteration": (K * ms_flow_step + ms_mle) / ms_iter}
print(json.dumps(res))
