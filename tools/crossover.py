"""Latency of the forward / backward (recomputing: bwd, from the stash: bwds) / reverse at several batch sizes on every
kernel family and arithmetic mode -> the crossovers the ABI's dispatch uses.  Columns: fp32 latency kernels forced,
throughput kernels forced in bf16x3 and fp32 mode, bf16x3 latency kernels forced (kernels that have no bf16x3 form
fall back to their fp32 one), throughput kernels forced in fp16x2 mode (forward and reverse; backward: as bf16x3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, lsnf_amd
dev = torch.device("cuda:0")
w = [t.to(dev) for t in bench.synth_weights(1)]
plan = lsnf_amd.prepare(w, 128, 64, 5)
def timeit(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for B in (32, 100, 256, 1024, 4096, 8192, 12288, 16384, 20480, 24576, 32768, 49152, 65536):
    z = torch.randn(B, 128, device=dev)
    res = []
    for name, thr, math in (("latency", 1 << 30, 0), ("throughput", 0, 1), ("throughput fp32", 0, 0), ("latency bf16x3", 1 << 30, 1),
                             ("throughput fp16x2", 0, 3)):
        lsnf_amd.flow.set_small_batch_max(thr)
        lsnf_amd.flow.set_math_mode(math)
        if which == "fwd":
            t = timeit(lambda: lsnf_amd.forward(plan, z), 300)
        elif which == "rev":
            t = timeit(lambda: lsnf_amd.reverse(plan, z), 100)
        else:   # "bwd": recomputing backward; "bwds": backward from the activation stash
            act = lsnf_amd.flow.new_act_saved(plan, B, dev) if which == "bwds" else None
            z1, ld, ll, sv = lsnf_amd.forward(plan, z, save_for_backward=True, act_saved=act)
            t = timeit(lambda: lsnf_amd.backward_z(plan, z1, sv, ll_scale=-1.0, act_saved=act), 100)
        res.append(t)
    print(f"{which} B={B:6d}  latency-kernel {res[0]:8.1f} us   throughput-kernel bf16x3 {res[1]:8.1f} us   fp32 MFMA {res[2]:8.1f} us   bf16x3 latency family forced (16-sample workgroups) {res[3]:8.1f} us   throughput-kernel fp16x2 (+ fix-up launch) {res[4]:8.1f} us")
