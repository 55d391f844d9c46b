"""Diagnostic: per-stage cycles and in-kernel clock of the latency forward (lsnf_small3_fwd_kernel<C, ST>) from the s_memtime /
s_memrealtime stamps of a -DLSNF_STAMPS build:
  make -C latent-space-normalizing-flow_amd/csrc BUILD=_build_stamps OUT=../_ablate/stamps.so EXTRA=-DLSNF_STAMPS
  LSNF_LIB_PATH=.../_ablate/stamps.so [LSNF_SMALL3_ST=2] python tools/stamps_small3.py [B]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
z = torch.randn(B, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
out = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
lib = lsnf_amd.load_library()
lib.lsnf_debug_stamps.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
lsnf_amd.flow.set_small_batch_max(1 << 30)
EXTRAS = len(sys.argv) > 2 and sys.argv[2] == "extras"          # the stash-writing forward (block outputs + activation stash)
act = lsnf_amd.flow.new_act_saved(plan, B, dev) if EXTRAS else None
saved = torch.empty(bench.DEPTH - 1, B, bench.NZ, device=dev) if EXTRAS else None
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 1.5:
    for _ in range(200):
        lsnf_amd.forward(plan, z, out=out, act_saved=act, z_saved_out=saved)
    torch.cuda.synchronize(); n += 200
buf = (ctypes.c_ulonglong * (2048 * 64))()
hip.hipMemcpy(buf, ctypes.c_void_p(lib.lsnf_debug_stamps()), ctypes.c_size_t(2048 * 64 * 8), 2)
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 64).astype(np.int64)
st = int(os.environ.get("LSNF_SMALL3_ST", "0"))
nw = min(2048, 4 * ((B + 16 * max(st, 1) - 1) // (16 * max(st, 1))))
s = s[:nw]
cyc = (s[:, 41] - s[:, 0]).astype(np.float64); rt = (s[:, 51] - s[:, 50]).astype(np.float64)
ok = rt > 0
med = lambda a: np.median(a[ok])
print(f"lsnf_small3_fwd_kernel{' (EXTRAS)' if EXTRAS else ''} ST={st or 'auto'} B={B} after {n} launches: in-kernel clock {med(cyc / np.maximum(rt, 1)) * 0.1:.3f} GHz; "
      f"wave lifetime {med(rt) / 100:.2f} us = {med(cyc):.0f} cycles")
print(f"  prologue (weight requests, row loads, split, barrier) {med(s[:, 1] - s[:, 0]):8.0f}")
for nm, (i, j) in [("    consts DMA issued", (0, 2)), ("    row loads issued", (2, 3)), ("    weight loads issued", (3, 4)), ("    rows waited for, split, LDS stores", (4, 5)), ("    barrier", (5, 1))]:
    print(f"  {nm:50s} {med(s[:, j] - s[:, i]):8.0f}")
print(f"  block 0                                              {med(s[:, 10] - s[:, 1]):8.0f}")
names = ["S1 k-tiles 0.. + the previous block's coupling", "barrier (mid-S1)", "S1 k-tiles HT.. + split v1 + barrier 1", "S2 + epilogue (relu, split)", "barrier 2",
         "S3 + epilogue", "barrier 3", "S4 MFMAs"]
idx = [(10, 11), (11, 12), (12, 13), (13, 14), (14, 15), (15, 16), (16, 17), (17, 18)]
for nm, (i, j) in zip(names, idx):
    if i == j: continue
    d = (s[:, j] - s[:, i])[ok]
    print(f"  block 1 {nm:44s} {np.median(d):8.0f}   (p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f})")
print(f"  block 1 total                                        {med(s[:, 20] - s[:, 10]):8.0f}")
print(f"  blocks 2..                                           {med(s[:, 40] - s[:, 20]):8.0f}")
print(f"  epilogue (stores, sums)                              {med(s[:, 41] - s[:, 40]):8.0f}")
