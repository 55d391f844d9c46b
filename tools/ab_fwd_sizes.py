"""GPU A/B of two builds of the library on the plain forward (+ in-kernel sums) at the strong-scaling shard sizes and the full size:
python tools/ab_fwd_sizes.py LIB_A LIB_B  (each size timed alternately, three rounds; a child process per library and round)."""
import json, os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
res = {}
for B in (65536, 32768, 16384, 8192):
    z = torch.randn(B, bench.NZ, device=dev)
    outs = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
    st = F.new_stats(dev)
    for _ in range(600): lsnf_amd.forward(plan, z, out=outs, stats=st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): lsnf_amd.forward(plan, z, out=outs, stats=st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 300 * 1e3)
    res[B] = sorted(ts)[2]
import json; print("RES" + json.dumps(res))
'''
libs = sys.argv[1:3]
acc = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LSNF_LIB_PATH=os.path.abspath(l)), capture_output=True, text=True).stdout
        acc[l].append(json.loads([x for x in out.splitlines() if x.startswith("RES")][0][3:]))
for l in libs:
    print(l)
    for B in ("65536", "32768", "16384", "8192"):
        v = sorted(r[B] for r in acc[l])
        print(f"   B={B:>6}: median {v[1]:7.2f} us  (min {v[0]:.2f}, max {v[2]:.2f})")
