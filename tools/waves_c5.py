"""GPU experiment: 4-wave (128-row, one wave per SIMD, 512 registers) vs 8-wave (256-row, 256 registers: spills at
f_width 128) workgroups for the throughput kernels at the CelebA-HQ geometry (nz=100, w=128), B = 65 536."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, %r)
import lsnf_amd
sys.path.insert(0, os.path.join(%r, "tools"))
dev = torch.device("cuda:0")
def timeit(fn, n=60, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nz, w in ((100, 128), (128, 64)):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    torch.manual_seed(1); np.random.seed(1)
    net = lsnf_amd._netF(hps, nz=nz)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if ".fc_zeros." in n_: p_.add_(0.05 * torch.randn_like(p_))
    net = net.to(dev); plan = net._plan()
    B = 65536
    z = torch.randn(B, nz, device=dev)
    for mode, nm in ((1, "bf16x3"), (3, "fp16x2")):
        lsnf_amd.flow.set_math_mode(mode)
        act = lsnf_amd.flow.new_act_saved(plan, B, dev)
        z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True, act_saved=act)
        r = {"fwd": timeit(lambda: lsnf_amd.forward(plan, z)), "fwd+stash": timeit(lambda: lsnf_amd.forward(plan, z, save_for_backward=True, act_saved=act)),
             "bwd(stash)": timeit(lambda: lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act)), "rev": timeit(lambda: lsnf_amd.reverse(plan, z))}
        print("nz=%%d w=%%d %%s waves=%%s: " %% (nz, w, nm, os.environ.get("LSNF_FORCE_WAVES", "auto")) + "  ".join("%%s %%.1f us" %% kv for kv in r.items()), flush=True)
''' % (ROOT, ROOT)
for wv in ("8", "4"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, LSNF_FORCE_WAVES=wv), capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
