"""GPU A/B: f_width 128 throughput forward as shipped (hipcc spills ~48 registers to scratch in the 8-wave instantiation)
against a -DLSNF_PARK_V build that parks v1 / v2 in the wave's own z_out rows instead (zero scratch), alternating in one
job.  Variant: hipcc ... -DLSNF_PARK_V -c lsnf_fwd3.hip / lsnf_fwd2h.hip, linked with the other objects into
latent-space-normalizing-flow_amd/_ablate/park.so.  Result (profiles/r02_ab_c5_park.txt): parked 186 us vs 153 us."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, %r)
import lsnf_amd
dev = torch.device("cuda:0")
def timeit(fn, n=300, warm=600):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=128, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=100).to(dev); plan = net._plan()
z = torch.randn(65536, 100, device=dev)
out = (torch.empty_like(z), torch.empty(65536, device=dev), torch.empty(65536, device=dev))
r = []
for mode, nm in ((1, "bf16x3"), (3, "fp16x2")):
    lsnf_amd.flow.set_math_mode(mode)
    r.append("%%s %%.1f us" %% (nm, timeit(lambda: lsnf_amd.forward(plan, z, out=out))))
print("  ".join(r))
''' % ROOT
for rep in range(2):
    for name, so in (("hipcc spills (shipped)", None), ("parked (-DLSNF_PARK_V)", os.path.join(ROOT, "latent-space-normalizing-flow_amd", "_ablate", "park.so"))):
        env = dict(os.environ)
        if so: env["LSNF_LIB_PATH"] = so
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=200)
        print(f"C5 nz=100 w=128 B=65536 forward, {name:24s}: {r.stdout.strip() or r.stderr[-300:]}", flush=True)
