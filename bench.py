#!/usr/bin/env python3
"""Headline benchmark: latent-samples/sec through flow + log-det (+ log-prob), CIFAR-10 flow geometry
nz=128 f_width=64 f_depth=5, B=65536 rows of synthetic z per evaluation (BASELINE.json configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1 without torchrun: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch: the fused forward launch (z -> z1, logdet, ll, and sum_b ll
accumulated in the kernel's epilogue, train.py:320) and -- for N > 1 -- the single RCCL all-reduce of that sum,
one per evaluation (asynchronous).  Steps are independent batches and rotate over `--streams` HIP streams; the
single-stream figure is reported beside it.  z, the prepared weights and all outputs are resident in HBM when the
timed region starts.

Arithmetic of the headline (`--math`, default bf16x3): every GEMM operand is split error-free into three bf16 terms
(24 significand bits, fp32's exponent range), six bf16 MFMAs per product, fp32 accumulation -- not narrower than the
reference's fp32 matmuls (model.py:187,326,347).  `--math fp32` runs the fp32-MFMA kernels; `--math fp16x2` (two fp16
terms, 22 bits) is an opt-in mode and never the headline.

N > 1: the headline is STRONG scaling -- the 65 536 rows of one evaluation sharded contiguously
over the ranks (`parallel.shard_bounds`), every evaluation all-reduced, `--strong-bucket` (8) evaluations per asynchronous collective -- and the same run also times WEAK scaling
(65 536 rows per GPU, `--weak-bucket` evaluations per collective) and reports it as `weak_scaling`.

Spread ("median and p10/p90"): the K-step timed region is repeated R = max(5, ceil(200 / K)) times behind ONE
clock ramp and ONE warm-up; `ms_per_step` is the MEDIAN of the R repetitions, `ms_per_step_p10` / `_p90` ride beside it (same
for the single-stream figure); the kernel-only loop of the roofline is timed per launch with HIP events and reports its median.

`--shard-of N` (one GPU): what ONE GPU of an N-GPU strong-scaling job does per step -- rows = 65 536 / N, same stream
rotation, in-kernel sums, the same reducer protocol -- next to the full-size step of the same run; prints the per-shard ms/step and
the N-GPU ceiling it implies (65 536 rows / shard time: no communication, no straggler), i.e. the bound on strong scaling that
one GPU decides.

Prints ONE JSON line on rank 0, including
  "roofline":     the forward kernel's ALGORITHMIC FLOP/s (327 680 FLOP/sample x rows / HIP-event time of a
                  single-stream kernel-only loop, measured in this run) against the dense peak of the matrix pipe it
                  runs on, plus the same rate against the fp32-MFMA peak and the executed-flops pipe utilisation;
  "cpu_baseline": the oracle (PyTorch-CPU restatement of the reference) timed on this box's host cores on a bounded
                  sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NZ, WIDTH, DEPTH = 128, 64, 5
B_GLOBAL = 65536
B_PER_GPU = B_GLOBAL            # (name used by tools/ for the single-GPU workload)
FLOP_PER_SAMPLE = DEPTH * (2 * NZ * NZ + 2 * (NZ // 2 * WIDTH + WIDTH * WIDTH + WIDTH * NZ))  # 327 680
BYTES_PER_SAMPLE_FUSED = 8 * NZ + 8                                                            # 1 032 (whole stack fused)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: "~2.5 PF dense" = 16 x the fp32 matrix rate (same table); fp16 = bf16
PEAK_HBM_GBS = 8000.0
# executed MFMA flops per algorithmic flop, pipe peak, kernel of the mode (B = 65 536 instantiation)
MODES = {
    "bf16x3": dict(exec_per_alg=6, peak=PEAK_BF16_MFMA_TFLOPS, pipe="bf16 MFMA", dtype="bf16x3",
                   kernel="lsnf_fwd3q_kernel<2, 8, 2>"),
    "fp32": dict(exec_per_alg=1, peak=PEAK_FP32_MFMA_TFLOPS, pipe="fp32 MFMA", dtype="f32",
                 kernel="lsnf_fwd_kernel<FwdCfg<2,2>, 8>"),
    "fp16x2": dict(exec_per_alg=3, peak=PEAK_BF16_MFMA_TFLOPS, pipe="fp16 MFMA", dtype="fp16x2",
                   kernel="lsnf_fwd2h_kernel<Fwd3Cfg<2,2>, 8>"),
}
MATH_TEXT = {
    "bf16x3": "bf16x3: every GEMM operand split error-free into three bf16 terms (3 x 8 = 24 significand bits, fp32's "
              "exponent range), six bf16 MFMAs (16x16x32) per product, fp32 accumulation; dropped terms <= 2^-26|w||x|; "
              "log-prob error vs float64 equals the fp32-MFMA kernel's (tests/test_gpu_forward.py::"
              "test_split_bf16_is_fp32_faithful)",
    "fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit an fmaf chain",
    "fp16x2": "fp16x2 (opt-in, narrower than the reference's fp32: 11+11 operand bits): two fp16 terms per operand, three "
              "fp16 MFMAs per product, range-guarded by a bf16x3 fix-up pass",
}


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
    z = torch.randn(B_GLOBAL, NZ, generator=torch.Generator().manual_seed(1234))
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
    return {"value": B_GLOBAL / med, "unit": "latent-samples/s", "cores": best, "kind": "port",
            "sample": f"{len(times)} full-size calls (B={B_GLOBAL}, nz={NZ}) of oracle.flow_log_prob, torch-CPU fp32 "
                      f"no_grad, median {med * 1e3:.1f} ms at {best} threads (probe ms/call: "
                      + ", ".join(f"{c}t={probe[c] * 1e3:.0f}" for c in cands) + f"; host has {all_threads} threads)"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`bench.py --gpus N` started without torch.distributed.run: start N fresh child ranks (one per GPU) BEFORE this
    process has made any GPU call, pass their output through and return the launcher's exit code.  The parent never
    touches the GPU (and never exec()s: it stays the parent of the ranks)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def selftest_ranks(args, rank, world):
    """`--selftest-launcher`: the multi-rank plumbing of this file on CPU tensors over gloo (no GPU, no kernels; the
    per-evaluation sums are synthetic) -- self-launch, env parsing, one all-reduce per evaluation through
    `PipelinedStatsReducer(bucket=1)`, barrier + max-over-ranks timing, one JSON line.  tests/test_bench_launcher.py."""
    import torch.distributed as dist
    from lsnf_amd import parallel
    dist.init_process_group("gloo")
    lo, hi = parallel.shard_bounds(B_GLOBAL, world, rank)
    red = parallel.PipelinedStatsReducer(torch.device("cpu"), bucket=1)
    dist.barrier()
    t0 = time.perf_counter()
    seen = []
    for it in range(args.steps):
        st = red.next_buffer()
        if it >= 2:
            seen.append(st[4:7].clone())              # evaluation it-2, reduced over the ranks
        st[4], st[5], st[6] = float(it + 1) * (rank + 1), -2.0 * (rank + 1), float(hi - lo)
        red.submit(st)
    fin = red.finish().clone()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    tri = world * (world + 1) / 2
    ok = (fin.tolist() == [args.steps * tri, -2.0 * tri, float(B_GLOBAL)]
          and all(s.tolist() == [(k + 1) * tri, -2.0 * tri, float(B_GLOBAL)] for k, s in enumerate(seen)))
    if rank == 0:
        print(json.dumps({"selftest": True, "ok": bool(ok), "n_gpus": world, "steps": args.steps, "scaling": "strong",
                          "rows_per_rank": hi - lo, "collectives_per_evaluation": 1, "elapsed_s": el.item()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def shard_mode(args, world, rank, dev, z_full, make_run, timed, summarize, ramp_out, reps, streams, main_stream, dtype):
    """`--shard-of N`: one GPU's share of an N-GPU strong-scaling step, and the ceiling it implies."""
    if world != 1:
        print("--shard-of runs on one GPU", file=sys.stderr)
        return 2
    n = args.shard_of
    if n < 1 or B_GLOBAL % n:
        print(f"--shard-of {n}: N must divide {B_GLOBAL}", file=sys.stderr)
        return 2
    rows = B_GLOBAL // n
    z = z_full[:rows].contiguous()
    bk = max(1, args.strong_bucket)            # the reducer protocol of the N-GPU run this stands for (no collective on one GPU)
    runs = {"shard": (make_run(z, bk, streams), make_run(z, bk, [main_stream])),
            "full": (make_run(z_full, 1, streams), make_run(z_full, 1, [main_stream]))}
    ramp_out(rows), ramp_out(B_GLOBAL)
    torch.cuda.synchronize()
    res = {}
    for name, (multi, single) in runs.items():
        r = B_GLOBAL // n if name == "shard" else B_GLOBAL
        res[name] = dict(summarize(timed(multi, args.steps, args.warmup), args.steps, r))
        res[name].update(summarize(timed(single, args.steps, args.warmup), args.steps, r, prefix="single_stream_"))
    sh, fu = res["shard"], res["full"]
    line = {
        "metric": f"latent-samples/sec through flow+logdet, nz=128: ONE GPU's shard of an {n}-GPU strong-scaling step (65536/{n} rows)",
        "value": sh["value"], "unit": "latent-samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sh["ms_per_step"], "ms_per_step_p10": sh["ms_per_step_p10"], "ms_per_step_p90": sh["ms_per_step_p90"],
        "us_per_step": sh["ms_per_step"] * 1e3, "timed_regions": reps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "shard_of": n, "rows": rows,
        "ms_per_step_single_stream": sh["single_stream_ms_per_step"],
        "ms_per_step_single_stream_p10": sh["single_stream_ms_per_step_p10"],
        "ms_per_step_single_stream_p90": sh["single_stream_ms_per_step_p90"],
        "full_size_ms_per_step": fu["ms_per_step"], "full_size_ms_per_step_single_stream": fu["single_stream_ms_per_step"],
        "full_size_value": fu["value"],
        # what N such GPUs would deliver with free communication and no straggler: 65 536 rows per shard time
        "implied_n_gpu_ceiling_samples_per_s": B_GLOBAL / (sh["ms_per_step"] * 1e-3),
        "implied_speedup_over_1_gpu": fu["ms_per_step"] / sh["ms_per_step"],
        "implied_speedup_over_1_gpu_single_stream": fu["single_stream_ms_per_step"] / sh["single_stream_ms_per_step"],
        "untimed_launches_before_timed_region": max(0, args.ramp) + args.warmup,
        "config": {"workload": f"CIFAR-10 flow prior nz=128 f_width=64 f_depth=5 affine, forward+logdet+log-prob with in-kernel "
                               f"sums, {rows} of the 65536 synthetic rows of one evaluation (the shard of GPU 0 of {n}); "
                               f"reducer with {bk} evaluation(s) per bank as in the N-GPU run, no collective (one GPU)",
                   "streams": len(streams), "clock_ramp_launches_before_warmup": max(0, args.ramp)},
    }
    print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--ramp", type=int, default=600,
                    help="untimed forward launches BEFORE the W warm-up steps that bring the chip to its steady clock (MI355X "
                         "needs ~50 ms of sustained load: 2.07 GHz in the first ~100 launches, 2.39 GHz afterwards); reported "
                         "in the JSON line as untimed_launches_before_timed_region = ramp + warmup; 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank logic on a box with fewer GPUs than ranks)")
    ap.add_argument("--math", choices=["bf16x3", "fp32", "fp16x2"], default="bf16x3",
                    help="arithmetic of the forward's GEMMs.  bf16x3 (default): error-free three-way bf16 split, 24 operand "
                         "bits, six bf16 MFMAs per product; fp32: fp32 MFMA; fp16x2: two-term fp16 split (22 operand bits: "
                         "narrower than the reference, opt-in only)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the independent steps rotate over (the load head of step i+1 overlaps the store tail of "
                         "step i); the single-stream ms/step is measured and reported as well")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: which form is the headline `value` (the other one is timed too and reported beside it)")
    ap.add_argument("--weak-bucket", type=int, default=32, help="evaluations per collective of the weak-scaling figure")
    ap.add_argument("--strong-bucket", type=int, default=8,
                    help="N > 1: evaluations per collective of the strong-scaling headline.  Every evaluation's [sum ll, sum logdet, rows] is "
                         "all-reduced; the pipelined reducer sends the sums of this many consecutive evaluations in ONE asynchronous collective "
                         "(the evaluations are independent, the sums are a few bytes).  With one collective per evaluation (1) the eager step "
                         "costs ~47 us of host time per rank against a ~20 us shard kernel at 8 GPUs (DESIGN.md section 6): host-bound")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="one GPU: time the shard of an N-GPU strong-scaling job (65 536 / N rows per step, same protocol) and "
                         "print the N-GPU ceiling it implies; the full-size step is timed in the same run")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="on: replay the K timed steps (forward launches, and for N > 1 their all-reduces, all on one stream then) from "
                         "ONE captured HIP graph instead of issuing them from Python.  With a live RCCL process group the eager step "
                         "costs ~31-50 us of host time (forward call 11, async all-reduce 21-24; tools/host_cost_allreduce.py) against a "
                         "~20 us shard kernel, so an N-GPU strong-scaling run issued from Python is HOST-bound; a captured graph of 20 "
                         "steps replays at 22 us per step in the one-rank rehearsal.  auto = off: collectives inside a captured graph "
                         "could only be rehearsed with one rank on this pool (a multi-stream capture around them crashed), so the "
                         "default stays the eager loop that has run on every backend")
    ap.add_argument("--reps", type=int, default=0, help="repetitions of the K-step timed region (default max(5, ceil(200 / K)))")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (before anything here has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_launcher:
        raise SystemExit(selftest_ranks(args, rank, world))
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
    from lsnf_amd import parallel
    MATH = {"fp32": lsnf_amd.flow.MATH_FP32, "bf16x3": lsnf_amd.flow.MATH_BF16X3, "fp16x2": lsnf_amd.flow.MATH_FP16X2}
    lsnf_amd.flow.set_math_mode(MATH[args.math])
    weights = synth_weights(1)
    plan = lsnf_amd.prepare([w.to(dev) for w in weights], NZ, WIDTH, DEPTH)
    n_streams = max(1, args.streams)
    main_stream = torch.cuda.current_stream()
    side_streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def fence(reducers, streams):
        for red, s in zip(reducers, streams):   # every outstanding all-reduce is complete before the clock is read
            with torch.cuda.stream(s):
                red.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_run(z, bucket, streams):
        """Buffers and reducers of one timed configuration -- allocated up front: an allocation between the clock ramp and
        the timed region idles the GPU for milliseconds and the chip drops back to its low clock."""
        rows = z.shape[0]
        outs = [(torch.empty_like(z), torch.empty(rows, device=dev), torch.empty(rows, device=dev)) for _ in streams]
        reducers = [parallel.PipelinedStatsReducer(dev, bucket=bucket) for _ in streams]
        # (the launch of a step with its buffers bound and checked once: ~4 us of host time per call instead of ~12)
        calls = [lsnf_amd.flow.BoundForward(plan, z, o) for o in outs]
        return z, outs, reducers, streams, calls

    reps = args.reps if args.reps > 0 else max(5, -(-200 // max(1, args.steps)))
    graph_used = []

    def timed(run, steps, warmup):
        """clock ramp (untimed, not steps) + W untimed steps, then `reps` x (K timed steps); returns the list of seconds per
        K-step region (max over ranks each)."""
        z, outs, reducers, streams, calls = run
        counter = [0]

        def step():
            # forward + log-prob; sum_b ll (train.py:320) is accumulated inside the kernel.  N > 1: the all-reduce of the
            # sums is submitted asynchronously on RCCL's stream and overlaps the following kernels (banks alternate)
            k = counter[0] % len(streams)
            counter[0] += 1
            red = reducers[k]
            if red.touches_stream():                # first / last evaluation of a bank: the reducer waits for / issues a collective
                with torch.cuda.stream(streams[k]):
                    stats = red.next_buffer()
                    calls[k](stats)                 # == lsnf_amd.forward(plan, z, out=outs[k], stats=stats)
                    red.submit(stats)
            else:                                   # in between: the launch names its stream, nothing else touches one
                stats = red.next_buffer()
                calls[k](stats, streams[k])
                red.submit(stats)

        # Clock ramp: MI355X needs ~50 ms of sustained load before it holds its steady shader clock (2.07 GHz in the first
        # ~100 launches, 2.39 GHz afterwards).  The metric is steady-state throughput, so the chip is brought there right
        # before the W warm-up steps, whatever W is; counted in untimed_launches_before_timed_region.  The ramp runs on the
        # main stream into a buffer of its own; the side streams wait for it.
        for _ in range(max(0, args.ramp)):
            lsnf_amd.forward(plan, z, out=ramp_out(z.shape[0]))
        for s_ in streams:
            if s_ is not main_stream:
                s_.wait_stream(main_stream)
        for _ in range(warmup):
            step()
        # The K timed steps as ONE captured graph (see --graph): forward launches and all-reduces on the same streams in the same
        # order as the eager loop; the reducers' flush + waits (what fence() does eagerly) are captured at the end.
        graph = None
        want = args.graph == "on"          # (auto = off: see --graph)
        if want and dist is not None and args.backend != "nccl":
            if rank == 0:
                print(f"[bench] --graph: the {args.backend} backend cannot be captured (host copies): issuing the steps eagerly", file=sys.stderr)
            want = False
        if want:
            fence(reducers, streams)
            try:
                cap = torch.cuda.Stream()
                # (the legacy default stream cannot take part in a capture: the single-stream run is captured on `cap` itself; with
                #  collectives in the graph everything is captured on ONE stream -- a fork / join over side streams around RCCL
                #  collectives crashed the process in the one-rank rehearsal, tools/host_cost_allreduce.py)
                if dist is not None:
                    streams, reducers = [cap], reducers[:1]
                    counter[0] = 0
                else:
                    streams[:] = [cap if s_ is main_stream else s_ for s_ in streams]
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(cap):
                    with torch.cuda.graph(g, stream=cap):
                        for s_ in streams:
                            if s_ is not cap:
                                s_.wait_stream(cap)
                        for _ in range(steps):
                            step()
                        for red, s_ in zip(reducers, streams):
                            with torch.cuda.stream(s_):
                                red.finish()
                        for s_ in streams:
                            if s_ is not cap:
                                cap.wait_stream(s_)
                torch.cuda.synchronize()
                g.replay()                                   # one untimed replay: the graph works before it is timed
                torch.cuda.synchronize()
                graph = g
            except Exception as e:                           # e.g. a backend that cannot be captured (gloo's host copies)
                if rank == 0:
                    print(f"[bench] graph capture not available ({type(e).__name__}: {str(e)[:120]}): issuing the steps eagerly", file=sys.stderr)
                torch.cuda.synchronize()
                graph = None
        graph_used.append(graph is not None)
        out = []
        for _ in range(reps):
            fence(reducers, streams) if graph is None else (torch.cuda.synchronize(), dist is not None and dist.barrier(), torch.cuda.synchronize())
            t0 = time.perf_counter()
            if graph is None:
                for _ in range(steps):
                    step()
                fence(reducers, streams)
            else:
                graph.replay()
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = t.item()
            out.append(elapsed)
        return out

    def pct(xs, q):
        xs = sorted(xs)
        return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]

    def summarize(els, steps, rows_all, prefix=""):
        med = pct(els, 0.5)
        return {prefix + "ms_per_step": med / steps * 1e3, prefix + "ms_per_step_p10": pct(els, 0.1) / steps * 1e3,
                prefix + "ms_per_step_p90": pct(els, 0.9) / steps * 1e3, prefix + "value": rows_all * steps / med}

    _ramp = {}

    def ramp_out(rows):
        if rows not in _ramp:
            _ramp[rows] = (torch.empty(rows, NZ, device=dev), torch.empty(rows, device=dev), torch.empty(rows, device=dev))
        return _ramp[rows]

    gen = torch.Generator().manual_seed(1234 + rank)
    z_weak = torch.randn(B_GLOBAL, NZ, generator=gen).to(dev)                  # 65 536 rows on every GPU
    if args.shard_of:
        raise SystemExit(shard_mode(args, world, rank, dev, z_weak, make_run, timed, summarize, ramp_out, reps,
                                    side_streams if n_streams > 1 else [main_stream], main_stream, MODES[args.math]["dtype"]))
    lo, hi = parallel.shard_bounds(B_GLOBAL, world, rank)
    z_strong = z_weak[: hi - lo].contiguous() if world > 1 else z_weak         # this rank's slab of ONE 65 536-row evaluation
    tmp = (torch.empty_like(z_weak), torch.empty(B_GLOBAL, device=dev), torch.empty(B_GLOBAL, device=dev))
    forms = ["strong", "weak"] if world > 1 else ["single"]
    runs = {}
    for form in forms:
        z = z_strong if form == "strong" else z_weak
        bucket = max(1, args.strong_bucket) if form != "weak" else max(1, args.weak_bucket)
        if world == 1:
            bucket = 1                                                         # (one GPU: no collective, the reducer only rotates its banks)
        runs[form] = (make_run(z, bucket, side_streams if n_streams > 1 else [main_stream]), make_run(z, bucket, [main_stream]), bucket)
    torch.cuda.synchronize()

    results = {}
    for form in forms:
        multi, single, bucket = runs[form]
        z = multi[0]
        ramp_out(z.shape[0])                          # (allocated before the ramp)
        els = timed(multi, args.steps, args.warmup)
        els1 = timed(single, args.steps, args.warmup)
        rows_all = B_GLOBAL if form != "weak" else world * B_GLOBAL
        results[form] = dict(summarize(els, args.steps, rows_all))
        one = summarize(els1, args.steps, rows_all)
        results[form].update({"ms_per_step_single_stream": one["ms_per_step"], "ms_per_step_single_stream_p10": one["ms_per_step_p10"],
                              "ms_per_step_single_stream_p90": one["ms_per_step_p90"], "value_single_stream": one["value"],
                              "timed_regions": reps, "rows_per_gpu": z.shape[0], "global_rows_per_step": rows_all,
                              "evaluations_per_collective": bucket if world > 1 else None})
    head = results["single"] if world == 1 else results[args.scaling]

    # kernel-only loop for the roofline: HIP events on the launch stream around back-to-back launches of one stream
    z1, logdet, ll = tmp

    def kernel_ms(mode, spread=None):
        """single-stream kernel-only loop of 200 back-to-back launches between two HIP events: ms per launch (what the
        rocprofv3 kernel trace of this command averages too).  spread: a second loop with an event after EVERY launch gives the
        per-launch median / p10 / p90 (each marker costs ~1 us of launch overlap, so those sit above the loop average)."""
        lsnf_amd.flow.set_math_mode(MATH[mode])
        for _ in range(10):
            lsnf_amd.forward(plan, z_weak, out=(z1, logdet, ll))
        torch.cuda.synchronize()
        kl = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(kl):
            lsnf_amd.forward(plan, z_weak, out=(z1, logdet, ll))
        e1.record()
        torch.cuda.synchronize()
        mean = e0.elapsed_time(e1) / kl
        if spread is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(kl + 1)]
            ev[0].record()
            for i in range(kl):
                lsnf_amd.forward(plan, z_weak, out=(z1, logdet, ll))
                ev[i + 1].record()
            torch.cuda.synchronize()
            ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(kl)]
            spread.update({"kernel_ms_per_launch_events_median": pct(ts, 0.5), "kernel_ms_per_launch_events_p10": pct(ts, 0.1),
                           "kernel_ms_per_launch_events_p90": pct(ts, 0.9), "launches_timed": kl})
        return mean

    others = {}
    for other in [m for m in ("fp32", "bf16x3", "fp16x2") if m != args.math]:   # the other arithmetic modes, kernel-only
        others[other] = {"kernel_ms": kernel_ms(other)}
        others[other]["ll"] = ll.clone()
    kern_spread = {}
    kern_ms = kernel_ms(args.math, kern_spread)   # also leaves the library in the measured mode
    flops = FLOP_PER_SAMPLE * B_GLOBAL
    for other in others:
        ll_o = others[other].pop("ll")
        o = others[other]
        o["samples_per_s_kernel_only"] = B_GLOBAL / (o["kernel_ms"] * 1e-3)
        o["algorithmic_tflops"] = flops / (o["kernel_ms"] * 1e-3) / 1e12
        o["frac_of_its_pipe_peak"] = o["algorithmic_tflops"] / MODES[other]["peak"]
        o["max_rel_ll_difference_to_measured_mode"] = ((ll - ll_o).abs() / ll.abs().clamp_min(1.0)).max().item()
        if other == "fp16x2":
            o["note"] = "opt-in mode, 22 operand bits: narrower than the reference's fp32 -- never the headline"
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
        mode = MODES[args.math]
        tflops = flops / (kern_ms * 1e-3) / 1e12          # algorithmic (fp32-equivalent) rate, measured in this run
        # committed profiler evidence of the same kernel (profiles/, newest round): CARRIED from files, not measured now
        carried = {"source": "carried from profiles/ (committed rocprofv3 runs of this command), not measured in this run"}
        try:
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_only_from_trace.json")))[-1]
            ent = json.load(open(tf)).get(mode["kernel"])
            if ent:
                us = ent["kernel_only_loop_last_200_avg_us"]
                carried["rocprofv3_kernel_trace"] = {"file": os.path.basename(tf), "avg_us": us,
                                                     "frac_from_trace": flops / (us * 1e-6) / 1e12 / mode["peak"]}
        except Exception:
            pass
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = (json.load(open(tfile)).get(args.math) or {}).get("hbm_bytes_per_launch")
                carried["traffic_file"] = "profiles/traffic.json (PMC passes FETCH_SIZE x 2 + WRITE_SIZE)"
            except Exception:
                traffic = None
        rl = {"bound": "mfma", "achieved": tflops, "peak": mode["peak"], "unit": "TFLOP/s", "frac": tflops / mode["peak"],
              "traffic": traffic, "kernel": mode["kernel"], "pipe": mode["pipe"],
              "kernel_ms": kern_ms, **kern_spread, "flop_per_launch": flops,
              "frac_vs_fp32_mfma_peak": tflops / PEAK_FP32_MFMA_TFLOPS, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
              "matrix_pipe_utilisation": mode["exec_per_alg"] * tflops / mode["peak"],
              "executed_mfma_flops_per_algorithmic_flop": mode["exec_per_alg"],
              "hbm_frac_secondary": BYTES_PER_SAMPLE_FUSED * B_GLOBAL / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
              "note": "achieved = ALGORITHMIC flops (327 680 per sample, SURVEY 8d) / HIP-event time per launch of a single-stream "
                      "kernel-only loop (200 back-to-back launches) of this run; kernel_ms_per_launch_events_*: the same loop with an "
                      "event after every launch (median / p10 / p90; the markers cost ~1 us each); frac = achieved / dense peak of the pipe the kernel runs on.  An fp32-"
                      "accurate product costs exec_per_alg MFMAs on that pipe, so frac <= 1/exec_per_alg; "
                      "matrix_pipe_utilisation prices the executed flops",
              "carried": carried}
        line = {
            "metric": "latent-samples/sec through flow+logdet, nz=128 B=65536",
            "value": head["value"], "unit": "latent-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "ms_per_step_p10": head["ms_per_step_p10"],
            "ms_per_step_p90": head["ms_per_step_p90"], "timed_regions": reps, "higher_is_better": True,
            # (strong at every N, N = 1 included: the 65 536 rows of one evaluation are what is fixed)
            "scaling": "strong" if world == 1 else args.scaling,
            "vs_baseline": None, "dtype": mode["dtype"], "data": "synthetic",
            "ms_per_step_single_stream": head["ms_per_step_single_stream"],
            "ms_per_step_single_stream_p10": head["ms_per_step_single_stream_p10"],
            "ms_per_step_single_stream_p90": head["ms_per_step_single_stream_p90"],
            "value_single_stream": head["value_single_stream"],
            "untimed_launches_before_timed_region": max(0, args.ramp) + args.warmup,
            "steps_issued_from": "one captured HIP graph per timed region (forward launches + all-reduces)" if (graph_used and graph_used[0]) else "Python, one call per launch",
            "config": {"workload": "CIFAR-10 flow prior nz=128 f_width=64 f_depth=5 affine, forward+logdet+log-prob, "
                                   "B=65536 synthetic z per evaluation (BASELINE.json configs[2])",
                       "rows_per_gpu": head["rows_per_gpu"], "global_rows": head["global_rows_per_step"],
                       "parallelism": (f"dp{world}, {args.scaling} scaling: rows sharded, weights replicated; sum ll / sum logdet / "
                                       f"rows of every evaluation all-reduced ({head['evaluations_per_collective']} "
                                       f"evaluation(s) per collective, backend {args.backend})") if world > 1 else "single GPU",
                       "streams": n_streams, "clock_ramp_launches_before_warmup": max(0, args.ramp),
                       "prepare_ms_not_in_step": prep_ms, "math": MATH_TEXT[args.math],
                       "other_math_modes": others},
            "roofline": rl,
        }
        if world > 1:
            other_form = "weak" if args.scaling == "strong" else "strong"
            line[other_form + "_scaling"] = results[other_form]
            line["ranks_seen_by_backend"] = dist.get_world_size()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
