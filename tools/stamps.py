"""Diagnostic: per-stage shader-clock stamps of the forward kernel (needs the -DLSNF_STAMPS build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
dev = torch.device("cuda:0")
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(bench.B_PER_GPU, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
out = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
for _ in range(int(os.environ.get("STAMP_WARM", "5"))):
    lsnf_amd.forward(plan, z, out=out)
torch.cuda.synchronize()
lib = lsnf_amd.load_library()
lib.lsnf_debug_stamps.restype = ctypes.c_void_p
ptr = lib.lsnf_debug_stamps()
n = 2048 * 64
buf = (ctypes.c_ulonglong * n)()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(n * 8), 2)
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 1, 64).astype(np.int64)
t0 = s[:, :, 0].min()
rel = s - t0
names = ["prologue(load z, consts, first acquire)"] + [f"blk{b} {st}" for b in range(5) for st in ("S1", "S2", "S3", "S4", "epilogue")] 
idx = [1] + [2 + 6 * b + k for b in range(5) for k in range(5)]
prev = rel[:, :, 0]
print(f"start skew over waves: max {int((rel[:,:,0]).max())} cycles")
tot = {}
for name, i in zip(names, idx):
    d = rel[:, :, i] - prev
    print(f"{name:42s} mean {d.mean():9.0f}  min {d.min():8d}  max {d.max():8d}")
    tot[name.split()[-1] if name.startswith('blk') else 'prologue'] = tot.get(name.split()[-1] if name.startswith('blk') else 'prologue', 0) + d.mean()
    prev = rel[:, :, i]
d = rel[:, :, 40] - prev; print(f"{'final stores issued':42s} mean {d.mean():9.0f}")
d = rel[:, :, 41] - rel[:, :, 40]; print(f"{'store drain (vmcnt 0)':42s} mean {d.mean():9.0f}  max {d.max()}")
print("totals per wave (cycles):", {k: int(v) for k, v in tot.items()}, "end:", int(rel[:, :, 41].mean()), "max end:", int(rel[:, :, 41].max()))
print("ideal MFMA cycles per wave if pipe shared by 2 waves: S1 32768, S2 8192, S3 8192, S4 16384 per block")
cyc = (s[:, :, 41] - s[:, :, 0]).astype(np.float64)
rt = (s[:, :, 51] - s[:, :, 50]).astype(np.float64)       # 100 MHz ticks
ok = rt > 0
print(f"in-kernel clock = d(s_memtime)/d(s_memrealtime)*100MHz: median {np.median(cyc[ok] / rt[ok]) * 0.1:.3f} GHz; "
      f"wave lifetime median {np.median(rt[ok]) / 100:.1f} us, {np.median(cyc[ok]):.0f} cycles")
st, en = s[:, :, 50].astype(np.float64) / 100.0, s[:, :, 51].astype(np.float64) / 100.0     # us (100 MHz realtime)
t_first = st.min()
print(f"kernel span by realtime stamps: first start 0, last end {en.max() - t_first:.1f} us")
for q in (0, 5, 25, 50, 75, 95, 100):
    print(f"  start p{q}: {np.percentile(st - t_first, q):7.1f} us    end p{q}: {np.percentile(en - t_first, q):7.1f} us")
wg_start = st.min(axis=1) - t_first
order = np.argsort(wg_start)
print("workgroups starting later than 5 us:", int((wg_start > 5).sum()), " later than 50 us:", int((wg_start > 50).sum()))
