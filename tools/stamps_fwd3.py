"""Diagnostic: in-kernel clock and wave lifetime of the forward kernels under sustained load, from the s_memtime /
s_memrealtime stamps of a -DLSNF_STAMPS build (make BUILD=_build_stamps OUT=../_ablate/stamps.so EXTRA=-DLSNF_STAMPS;
LSNF_LIB_PATH=.../_ablate/stamps.so python tools/stamps_fwd3.py).  Runs >= 2 s of back-to-back launches on random data."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(bench.B_PER_GPU, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
out = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
lib = lsnf_amd.load_library()
lib.lsnf_debug_stamps.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
for mode, name in ((lsnf_amd.flow.MATH_BF16X3, "bf16x3 16x16x32 (lsnf_fwd3b_kernel)"), (lsnf_amd.flow._MATH_X_BF16X3_32, "bf16x3 32x32x16 (lsnf_fwd3_kernel)"),
                   (lsnf_amd.flow.MATH_FP32, "fp32 MFMA (lsnf_fwd_kernel)")):
    lsnf_amd.flow.set_math_mode(mode)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(200):
            lsnf_amd.forward(plan, z, out=out)
        torch.cuda.synchronize(); n += 200
    buf = (ctypes.c_ulonglong * (2048 * 64))()
    hip.hipMemcpy(buf, ctypes.c_void_p(lib.lsnf_debug_stamps()), ctypes.c_size_t(2048 * 64 * 8), 2)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 64).astype(np.int64)
    cyc = (s[:, 41] - s[:, 0]).astype(np.float64)
    rt = (s[:, 51] - s[:, 50]).astype(np.float64)
    ok = rt > 0
    print(f"{name}: after {n} launches: in-kernel clock median {np.median(cyc[ok] / rt[ok]) * 0.1:.3f} GHz "
          f"(p10 {np.percentile(cyc[ok] / rt[ok], 10) * 0.1:.3f}, p90 {np.percentile(cyc[ok] / rt[ok], 90) * 0.1:.3f}); "
          f"wave lifetime median {np.median(rt[ok]) / 100:.1f} us = {np.median(cyc[ok]):.0f} cycles")
