"""Diagnostic: in-kernel clock, wave lifetime and per-phase cycles of the software-pipelined bf16x3 forward
(lsnf_fwd3p.hip) under sustained load, from the s_memtime / s_memrealtime stamps of a -DLSNF_STAMPS build
(make -C latent-space-normalizing-flow_amd/csrc BUILD=_build_stamps OUT=../_ablate/stamps.so EXTRA=-DLSNF_STAMPS;
LSNF_LIB_PATH=.../_ablate/stamps.so python tools/stamps_fwd3p.py)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
BROWS = int(sys.argv[2]) if len(sys.argv) > 2 else bench.B_GLOBAL
z = torch.randn(BROWS, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
out = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
lib = lsnf_amd.load_library()
lib.lsnf_debug_stamps.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
which = sys.argv[1] if len(sys.argv) > 1 else "q"          # p: lsnf_fwd3p_kernel (32x32x16), q: lsnf_fwd3q_kernel (16x16x32)
lsnf_amd.flow.set_math_mode(lsnf_amd.flow.MATH_BF16X3 if which == "q" else lsnf_amd.flow._MATH_X_BF16X3_PIPE)
lsnf_amd.flow.set_small_batch_max(0)
stash = len(sys.argv) > 3 and sys.argv[3] == "stash"         # the stash-writing instantiation (lsnf_fwd3q_kernel<.., STASH>)
act = lsnf_amd.flow.new_act_saved(plan, BROWS, dev) if stash else None
saved = torch.empty(bench.DEPTH - 1, BROWS, bench.NZ, device=dev) if stash else None
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(200):
        lsnf_amd.forward(plan, z, out=out, act_saved=act, z_saved_out=saved)
    torch.cuda.synchronize(); n += 200
buf = (ctypes.c_ulonglong * (2048 * 64))()
hip.hipMemcpy(buf, ctypes.c_void_p(lib.lsnf_debug_stamps()), ctypes.c_size_t(2048 * 64 * 8), 2)
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 64).astype(np.int64)
cyc = (s[:, 41] - s[:, 0]).astype(np.float64); rt = (s[:, 51] - s[:, 50]).astype(np.float64)
ok = rt > 0
tag = " with stash" if stash else ""
print(f"lsnf_fwd3{which}_kernel{tag} after {n} launches: in-kernel clock median {np.median(cyc[ok] / rt[ok]) * 0.1:.3f} GHz; wave lifetime "
      f"median {np.median(rt[ok]) / 100:.1f} us = {np.median(cyc[ok]):.0f} cycles")
print(f"  prologue (row loads, first split, barrier 0) {np.median((s[:, 1] - s[:, 0])[ok]):9.0f} cycles")
print(f"  block 0                               {np.median((s[:, 10] - s[:, 1])[ok]):9.0f}")
for half, sel in (("early half (waves 0-3)", np.arange(2048) % 8 < 4), ("late half (waves 4-7)", np.arange(2048) % 8 >= 4)):
    print(f"  {half}:")
    for i, nm in enumerate(["S1a (96 MFMA)", "S1b (96)", "S2+S3 (96) + last-block branch", "S4 (96) + coupling"]):
        d = (s[:, 11 + i] - s[:, 10 + i])[ok & sel]
        print(f"    block 1 {nm:30s} {np.median(d):9.0f}   (p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f})")
    print(f"    block 1 total                         {np.median((s[:, 14] - s[:, 10])[ok & sel]):9.0f}   (384 MFMA x 32 cycles x 2 waves per SIMD = 24576)")
print(f"  blocks 2..4                           {np.median((s[:, 40] - s[:, 14])[ok]):9.0f}")
print(f"  epilogue (stores, sums)               {np.median((s[:, 41] - s[:, 40])[ok]):9.0f}")
