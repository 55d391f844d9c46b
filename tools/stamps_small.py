"""Diagnostic: per-stage shader-clock stamps of the latency (small-batch) forward kernel; needs the -DLSNF_STAMPS
build (make BUILD=_build_stamps OUT=../liblsnf_flow_stamps.so EXTRA=-DLSNF_STAMPS; LSNF_LIB_PATH=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, lsnf_amd
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(B, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
for _ in range(50):
    lsnf_amd.forward(plan, z)
torch.cuda.synchronize()
lib = lsnf_amd.load_library()
lib.lsnf_debug_stamps.restype = ctypes.c_void_p
ptr = lib.lsnf_debug_stamps()
nwg = (B + 31) // 32
n = 2048 * 64
buf = (ctypes.c_ulonglong * n)()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(n * 8), 2)
s = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 64).astype(np.int64)[:nwg]
names = {0: "start", 1: "prologue done"}
for b in range(bench.DEPTH):
    for k, nm in enumerate(["S1 work", "S1 barrier", "S2 work", "S2 barrier", "S3 work", "S3 barrier", "S4 work", "S4 barrier",
                            "coupling work", "coupling barrier"]):
        names[2 + 10 * b + k] = f"blk{b} {nm}"
names[60] = "epilogue"
order = sorted(names)
for wg in range(min(nwg, 2)):
    print(f"--- workgroup {wg}: cycles spent per segment, waves 0..3")
    prev = s[wg, :, 0]
    for i in order[1:]:
        d = s[wg, :, i] - prev
        print(f"{names[i]:24s} " + " ".join(f"{int(x):7d}" for x in d))
        prev = s[wg, :, i]
    print(f"{'total':24s} " + " ".join(f"{int(x):7d}" for x in (s[wg, :, 60] - s[wg, :, 0])))
agg = {}
prev = s[:, :, 0]
for i in order[1:]:
    d = (s[:, :, i] - prev).max(axis=1).mean()
    key = names[i].split(" ", 1)[1] if names[i].startswith("blk") else names[i]
    agg[key] = agg.get(key, 0) + d
    prev = s[:, :, i]
print("sum over blocks of (max over waves), mean over workgroups, cycles:")
for k, v in agg.items():
    print(f"  {k:20s} {v:9.0f}")
