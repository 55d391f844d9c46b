"""Times the split-bf16 forward of every variant built by tools/ablate_fwd3.sh (one subprocess per library)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch, bench, lsnf_amd
    dev = torch.device("cuda:0")
    plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
    z = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
    lsnf_amd.flow.set_math_mode(1)
    out = (torch.empty_like(z), torch.empty(65536, device=dev), torch.empty(65536, device=dev))
    for _ in range(800): lsnf_amd.forward(plan, z, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): lsnf_amd.forward(plan, z, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ['LSNF_LIB_PATH']):28s} {e0.elapsed_time(e1) / 300 * 1e3:7.1f} us", flush=True)
else:
    for lib in sorted(glob.glob(os.path.join(ROOT, "latent-space-normalizing-flow_amd", "_ablate", "f3_*.so"))):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, LSNF_LIB_PATH=lib), check=False)
