"""GPU: is the bf16x3 forward power-limited?  Times the kernels on random z (the workload) and on all-zero z (same
instruction stream, almost no operand toggling in the MFMAs): if the zero run is much faster, the chip is holding its
clock down under the data-dependent power of the matrix pipe, and time follows energy, not cycle count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, lsnf_amd
dev = torch.device("cuda:0")
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
lsnf_amd.flow.set_small_batch_max(0)


def t_us(z, mode):
    lsnf_amd.flow.set_math_mode(mode)
    outs = (torch.empty_like(z), torch.empty(z.shape[0], device=dev), torch.empty(z.shape[0], device=dev))
    for _ in range(1500):
        lsnf_amd.forward(plan, z, out=outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        lsnf_amd.forward(plan, z, out=outs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 500 * 1e3


zr = torch.randn(65536, bench.NZ, generator=torch.Generator().manual_seed(1234)).to(dev)
zz = torch.zeros_like(zr)
F = lsnf_amd.flow
for name, mode in (("lsnf_fwd3q_kernel (16x16x32, pipelined)", F.MATH_BF16X3), ("lsnf_fwd3p_kernel (32x32x16, pipelined)", F._MATH_X_BF16X3_PIPE),
                   ("lsnf_fwd3b_kernel (16x16x32, phases)", F.MATH_BF16X3_PHASED), ("lsnf_fwd3_kernel (32x32x16, phases)", F._MATH_X_BF16X3_32),
                   ("lsnf_fwd2h_kernel (fp16x2)", F.MATH_FP16X2), ("lsnf_fwd_kernel (fp32 MFMA)", F.MATH_FP32)):
    a, b = t_us(zr, mode), t_us(zz, mode)
    print(f"{name:44s} random z {a:7.2f} us   zero z {b:7.2f} us   ratio {a / b:.3f}", flush=True)
