"""Loop the flow-MLE step (forward + parameter backward, train.py:404-415) at the reference's batch size so that
`rocprofv3 --kernel-trace --stats -- python tools/run_mle.py` shows which kernels its time goes to."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=100).to(dev)
z = torch.randn(B, 100, device=dev)
obj = torch.zeros(B, device=dev)

def mle():
    net.zero_grad(set_to_none=True)
    a, b, _ = net(z, objective=obj)
    (-(-0.5 * (a ** 2).sum(1) + 1.8378770664093453 + b).mean()).backward()

if len(sys.argv) > 3 and sys.argv[3] == "fused":
    def mle():
        net.mle_grads(z, max_norm=100.0, reuse_buffers=True)

for _ in range(20): mle()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): mle()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"mle step B={B}: {(t1 - t0) / n * 1e6:.1f} us wall per step")
