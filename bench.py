#!/usr/bin/env python3
"""Headline benchmark: latent-samples/sec through flow + log-det (+ log-prob), CIFAR-10 flow
geometry nz=128 f_width=64 f_depth=5, B=65536 rows PER GPU of synthetic z (BASELINE.json configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch: the fused forward launch (z -> z1, logdet, ll, and
sum_b ll accumulated in the kernel's epilogue, train.py:320) and -- for N > 1 -- the single RCCL all-reduce
of that sum (asynchronous).  Steps are independent batches and alternate over two HIP streams (--streams).
z, the prepared weights and all outputs are resident in HBM when the timed region starts.
Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the forward kernel's algorithmic FLOP/s (HIP-event timed, kernel-only loop) against
                  the dense fp32 MFMA peak of MI355X,
  "cpu_baseline": the oracle (PyTorch-CPU restatement of the reference) timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NZ, WIDTH, DEPTH = 128, 64, 5
B_PER_GPU = 65536
FLOP_PER_SAMPLE = DEPTH * (2 * NZ * NZ + 2 * (NZ // 2 * WIDTH + WIDTH * WIDTH + WIDTH * NZ))  # 327 680
BYTES_PER_SAMPLE_FUSED = 8 * NZ + 8                                                            # 1 032 (whole stack fused)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: "~2.5 PF dense" = 16 x the fp32 matrix rate (same table)
SPLIT_MFMA_PER_PRODUCT = 6      # bf16x3 mode: six bf16 MFMAs of K=16 carry one fp32-accurate 32x32x16 product
PEAK_HBM_GBS = 8000.0


def synth_weights(seed=1):
    """Reference-style init (orthogonal W, 0.05 N(0,1) elsewhere) + 0.05 N(0,1) on fc_zeros, built
    with torch/numpy RNG only (no oracle import on the product path)."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    half = NZ // 2
    out = []
    for _ in range(DEPTH):
        rn = lambda *s: torch.randn(*s, generator=g) * 0.05  # noqa: E731
        q = torch.tensor(np.linalg.qr(rs.randn(NZ, NZ))[0], dtype=torch.float32)
        out += [rn(NZ), rn(NZ), q, rn(half, WIDTH), rn(WIDTH), rn(WIDTH), rn(WIDTH, WIDTH), rn(WIDTH), rn(WIDTH),
                rn(WIDTH, NZ), rn(NZ), rn(NZ)]
    return out


def cpu_baseline(weights, budget_s=14.0):
    """Oracle timed on the host cores.  Bounded sample: full-size (65536-row) forward+log-prob calls.  The thread
    count is probed first (one call each at 8/16/32/64/all threads: eager PyTorch-CPU ops of this size do not scale to
    every core of a big host) and the fastest setting is then timed for the rest of the budget."""
    from oracle import flow_oracle as O
    import lsnf_amd
    keys = lsnf_amd.flow.BLOCK_PARAM_KEYS
    p = {}
    for i in range(DEPTH):
        for j, k in enumerate(keys):
            t = weights[i * 12 + j]
            p[O.block_prefix(i) + k] = t.reshape(1, -1) if t.dim() == 1 else t
    z = torch.randn(B_PER_GPU, NZ, generator=torch.Generator().manual_seed(1234))
    all_threads = torch.get_num_threads()
    cands = sorted({c for c in (8, 16, 32, 64, all_threads) if c <= all_threads})
    t_start = time.perf_counter()
    probe = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            O.flow_log_prob(p, z)                       # warm-up at this setting
            t0 = time.perf_counter()
            O.flow_log_prob(p, z)
            probe[c] = time.perf_counter() - t0
        best = min(probe, key=probe.get)
        torch.set_num_threads(best)
        times = []
        while (time.perf_counter() - t_start < budget_s) or len(times) < 3:
            t0 = time.perf_counter()
            O.flow_log_prob(p, z)
            times.append(time.perf_counter() - t0)
    torch.set_num_threads(all_threads)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B_PER_GPU / med, "unit": "latent-samples/s", "cores": best, "kind": "port",
            "sample": f"{len(times)} full-size calls (B={B_PER_GPU}, nz={NZ}) of oracle.flow_log_prob, torch-CPU fp32 "
                      f"no_grad, median {med * 1e3:.1f} ms at {best} threads (probe ms/call: "
                      + ", ".join(f"{c}t={probe[c] * 1e3:.0f}" for c in cands) + f"; host has {all_threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)   # the chip needs ~50 ms of sustained load to reach its steady clock
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank logic on a box with fewer GPUs than ranks)")
    ap.add_argument("--math", choices=["fp16x2", "bf16x3", "fp32"], default="fp16x2",
                    help="arithmetic of the forward's GEMMs: fp16x2 = two-term fp16 split, three fp16 MFMAs per product, "
                         "range-guarded by a bf16x3 fix-up pass (fp32-class accuracy, the library default); bf16x3 = "
                         "error-free three-way bf16 split, six bf16 MFMAs per product; fp32 = fp32 MFMA")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the independent steps rotate over (the load head of step i+1 and the early-exit fix-up "
                         "launch of step i overlap the store tail of step i; measured 1: 75.8, 2: 70.3, 3: 60.7, 4: 64.7, 6: 61.1 us/step)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the flow path has no CPU fallback")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL
        else:
            dist.init_process_group(args.backend)

    import lsnf_amd
    MATH = {"fp32": lsnf_amd.flow.MATH_FP32, "bf16x3": lsnf_amd.flow.MATH_BF16X3, "fp16x2": lsnf_amd.flow.MATH_FP16X2}
    lsnf_amd.flow.set_math_mode(MATH[args.math])
    weights = synth_weights(1)
    plan = lsnf_amd.prepare([w.to(dev) for w in weights], NZ, WIDTH, DEPTH)
    z = torch.randn(B_PER_GPU, NZ, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    from lsnf_amd import parallel

    # Steps are independent batches (synthetic z), so consecutive steps alternate over `--streams` HIP streams, each
    # with its own output buffers: while the last workgroups of step i drain their stores, the workgroups of step
    # i+1 already load their rows on the CUs that have become free (fp32 generation: 174 -> 164 us per step; today's
    # fp16x2 forward, whose every launch is followed by an early-exit fix-up launch: 70.3 us at 2 streams, 60.7 at 3).  Every step is
    # still a complete forward over 65 536 rows; K steps are timed, as the contract says.
    n_streams = max(1, args.streams)
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [main_stream]
    outs = [(torch.empty_like(z), torch.empty(B_PER_GPU, device=dev), torch.empty(B_PER_GPU, device=dev))
            for _ in range(n_streams)]
    z1, logdet, ll = outs[0]
    # one collective per REDUCE_BUCKET evaluations of a stream (each evaluation's sums travel, bucketed): the forward
    # kernels fill the chip exactly (one workgroup per CU), so every collective kernel delays one of their workgroups
    REDUCE_BUCKET = 32
    reducers = [parallel.PipelinedStatsReducer(dev, bucket=REDUCE_BUCKET) for _ in range(n_streams)]
    counter = [0]
    torch.cuda.synchronize()

    def step():
        # forward + log-prob; sum_b ll (train.py:320) is accumulated inside the kernel.  N > 1: the single all-reduce
        # of that sum is submitted asynchronously and overlaps the following kernels (stats buffers alternate).
        k = counter[0] % n_streams
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            stats = reducers[k].next_buffer()
            lsnf_amd.forward(plan, z, out=outs[k], stats=stats)
            reducers[k].submit(stats)

    def fence():
        for k in range(n_streams):  # every outstanding all-reduce is complete before the clock is read
            with torch.cuda.stream(streams[k]):
                reducers[k].finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock ramp (not steps, not timed): MI355X needs ~50 ms of sustained load before it holds its steady shader clock
    # (2.07 GHz in the first ~100 launches, 2.39 GHz afterwards).  The metric is steady-state
    # throughput, so the chip is brought to that state first, whatever W the caller passes.
    RAMP_LAUNCHES = 600
    for _ in range(RAMP_LAUNCHES):
        lsnf_amd.forward(plan, z, out=outs[0])
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # kernel-only loop for the roofline: HIP events on the launch stream around K back-to-back launches
    def kernel_ms(mode):
        lsnf_amd.flow.set_math_mode(MATH[mode])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            lsnf_amd.forward(plan, z, out=(z1, logdet, ll))
        torch.cuda.synchronize()
        kl = max(20, min(args.steps, 200))
        e0.record()
        for _ in range(kl):
            lsnf_amd.forward(plan, z, out=(z1, logdet, ll))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / kl
    others = {}
    for other in [m for m in ("fp32", "bf16x3", "fp16x2") if m != args.math]:   # the other arithmetic modes, reported beside the measured one
        others[other] = {"kernel_ms": kernel_ms(other)}
        others[other]["ll"] = ll.clone()
    kern_ms = kernel_ms(args.math)            # also leaves the library in the measured mode
    for other in others:
        ll_o = others[other].pop("ll")
        others[other]["samples_per_s_kernel_only"] = B_PER_GPU / (others[other]["kernel_ms"] * 1e-3)
        others[other]["max_rel_ll_difference_to_measured_mode"] = ((ll - ll_o).abs() / ll.abs().clamp_min(1.0)).max().item()
    # prepare (weight folding + fp64 Gauss-Jordan), amortised over the Langevin loop in production
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wd = [w.to(dev) for w in weights]
    p0.record()
    for _ in range(5):
        lsnf_amd.prepare(wd, NZ, WIDTH, DEPTH, plan=plan)
    p1.record()
    torch.cuda.synchronize()
    prep_ms = p0.elapsed_time(p1) / 5

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B_PER_GPU * args.steps / elapsed
        tflops = FLOP_PER_SAMPLE * B_PER_GPU / (kern_ms * 1e-3) / 1e12          # algorithmic (fp32-equivalent) rate
        if args.math == "fp16x2":   # 3 fp16 MFMA flops per algorithmic flop, priced against the dense fp16 peak (= the bf16 one)
            rl = {"bound": "mfma", "achieved": 3 * tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": 3 * tflops / PEAK_BF16_MFMA_TFLOPS, "kernel": "lsnf_fwd2h_kernel<Fwd3Cfg<2,2>, 8>",
                  "note": "executed fp16 MFMA flops (3 per algorithmic flop: operands split into two fp16 terms, products "
                          "w1x1 + w1x2 + w2x1) vs the dense fp16 peak; algorithmic_tflops is the fp32-equivalent rate, 1.0 of "
                          "the fp32 MFMA peak would be 157.3.  kernel_ms is measured around the launch PAIR the mode issues "
                          "(the fp16 kernel and the bf16x3 fix-up pass behind it, which exits at once unless an operand left "
                          "fp16's range: ~2 us).  The matrix pipe is busy about half of the kernel: the rest is the operand "
                          "split (2 VALU per element), the coupling epilogue and the z rows in / out, none of which "
                          "overlap MFMA issue on a SIMD (DESIGN.md section 5)"}
        elif args.math == "bf16x3":   # the matrix pipe executes 6 bf16 MFMA flops per algorithmic flop: price THAT against the bf16 peak
            rl = {"bound": "mfma", "achieved": SPLIT_MFMA_PER_PRODUCT * tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": SPLIT_MFMA_PER_PRODUCT * tflops / PEAK_BF16_MFMA_TFLOPS, "kernel": "lsnf_fwd3b_kernel<Fwd3Cfg<2,2>, 8>",
                  "note": "executed bf16 MFMA flops (6 per algorithmic flop) vs the dense bf16 peak; algorithmic_tflops is "
                          "the fp32-equivalent rate, 1.0 of the fp32 MFMA peak would be 157.3.  Calibration (tools/micro/"
                          "mfma_bf16_shapes.hip, DESIGN.md section 5): a bare loop of this MFMA shape (16x16x32) sustains 1819 "
                          "TFLOP/s on random operands (the chip gives clock back under the bf16 pipe: this kernel runs at "
                          "2.0-2.2 GHz), 2234 on zeros",
                  "bare_mfma_loop_on_random_operands_tflops": 1819.0}
        else:
            rl = {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": tflops / PEAK_FP32_MFMA_TFLOPS, "kernel": "lsnf_fwd_kernel<FwdCfg<2,2>, 8>"}
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = (json.load(open(tfile)).get(args.math) or {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the committed rocprofv3 kernel-trace average of the same kernel in the same command (profiles/, newest round),
        # next to the live HIP-event figure above
        trace_us = None
        try:
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_only_from_trace.json")))[-1]
            ent = json.load(open(tf)).get(rl["kernel"])
            if ent:
                trace_us = {"file": os.path.basename(tf), "avg_us": ent["kernel_only_loop_last_200_avg_us"]}
        except Exception:
            trace_us = None
        rl["rocprofv3_kernel_trace"] = trace_us
        line = {
            "metric": "latent-samples/sec through flow+logdet, nz=128 B=65536",
            "value": value, "unit": "latent-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp16x2": "fp16x2", "bf16x3": "bf16x3", "fp32": "f32"}[args.math], "data": "synthetic",
            "config": {"workload": "CIFAR-10 flow prior nz=128 f_width=64 f_depth=5 affine, forward+logdet+log-prob, "
                                   "B=65536 synthetic z per GPU (BASELINE.json configs[2])",
                       "rows_per_gpu": B_PER_GPU, "global_rows": world * B_PER_GPU,
                       "parallelism": (f"dp{world} (rows sharded; sum ll / sum logdet / rows of every evaluation all-reduced, "
                                       f"{REDUCE_BUCKET} evaluations per collective)") if world > 1 else "single GPU",
                       "streams": n_streams, "clock_ramp_launches_before_warmup": RAMP_LAUNCHES,
                       "prepare_ms_not_in_step": prep_ms,
                       "math": {"fp16x2": "fp16x2: every GEMM operand split into two fp16 terms (11+11 significand bits), three fp16 "
                                          "MFMAs (16x16x32) per product, fp32 accumulation; dropped terms <= 2^-22|w||x|, log-prob "
                                          "error vs float64 equals the fp32-MFMA kernel's (tests/test_gpu_forward.py::"
                                          "test_split_bf16_is_fp32_faithful, ::test_split_bf16_dynamic_range); operands outside "
                                          "fp16's range are caught in-kernel and the launch recomputed by the bf16x3 kernel "
                                          "(::test_fp16_split_range_guard)",
                                "bf16x3": "bf16x3: every GEMM operand split error-free into three bf16 terms, six bf16 MFMAs "
                                          "(16x16x32) per product, fp32 accumulation; log-prob error vs float64 equals the "
                                          "fp32-MFMA kernel's (tests/test_gpu_forward.py::test_split_bf16_is_fp32_faithful)",
                                "fp32": "fp32 MFMA"}[args.math],
                       "other_math_modes": others},
            "roofline": dict(rl, traffic=traffic, kernel_ms=kern_ms, flop_per_launch=FLOP_PER_SAMPLE * B_PER_GPU,
                             algorithmic_tflops=tflops, fp32_mfma_peak=PEAK_FP32_MFMA_TFLOPS,
                             hbm_frac_secondary=BYTES_PER_SAMPLE_FUSED * B_PER_GPU / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
