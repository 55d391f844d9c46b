"""GPU: time the ablation builds of tools/ablate_fwd3p.sh (one subprocess per build, B = 65 536, kernel-only loop)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
import bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
lsnf_amd.flow.set_math_mode(lsnf_amd.flow.MATH_BF16X3)
z = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
outs = (torch.empty_like(z), torch.empty(65536, device=dev), torch.empty(65536, device=dev))
for _ in range(1500): lsnf_amd.forward(plan, z, out=outs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500): lsnf_amd.forward(plan, z, out=outs)
e1.record(); torch.cuda.synchronize()
print("%%8.2f us" %% (e0.elapsed_time(e1) / 500 * 1e3))
''' % ROOT
for so in sorted(glob.glob(os.path.join(ROOT, "latent-space-normalizing-flow_amd", "_ablate", "p_*.so"))):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, LSNF_LIB_PATH=so), capture_output=True, text=True, timeout=120)
    print(f"{os.path.basename(so):40s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
