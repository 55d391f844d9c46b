#!/usr/bin/env python3
"""Synthetic training iterations of the reference's loop (train.py:376-415) on top of the MI355X flow prior:
K-step Langevin sampling (generator gradient by torch autograd + the fused flow step), generator Adam step,
flow-MLE Adam step -- for the four dataset geometries of the reference's README, with synthetic images.

    python examples/train_synthetic.py --dataset svhn                               # BASELINE.json configs[1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           examples/train_synthetic.py --dataset celeba                             # configs[3]: data parallel
    ... --dataset celeba_hq256                                                      # configs[4]

Multi-GPU: one process per GPU (RCCL); the image batch and the latents are sharded by rank, the generator is wrapped
in stock DistributedDataParallel, the flow's parameter gradients travel as ONE flat bucket
(`lsnf_amd.parallel.allreduce_gradients`), its weights are broadcast once.  The Langevin loop itself needs no
communication (per-sample).  The generator is `lsnf_amd._netG`, the table-driven mirror of the reference's `_netG`
(model.py:48-157: stock ConvTranspose2d stacks, i.e. MIOpen), `--no-tune` leaves MIOpen's find mode and channels-last off.
Prints one JSON line per run (rank 0): ms/iteration and the flow's share of it."""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn

import lsnf_amd
from lsnf_amd import langevin, parallel

# dataset -> (image size, nz, ngf, f_width, Langevin steps): README.md:30-66 of the reference
GEOMETRY = {"svhn": (32, 100, 64, 64, 20), "cifar10": (32, 128, 128, 64, 40),
            "celeba": (64, 100, 128, 64, 20), "celeba_hq256": (256, 100, 128, 128, 20)}


REF_DATASET = {"svhn": "svhn", "cifar10": "cifar10", "celeba": "celeba_crop", "celeba_hq256": "celeba_hq256"}


def make_generator(dataset, nz, ngf, nc, tune):
    """The reference's generator for this dataset (train.py:266-267: `_netG(args)` + xavier init)."""
    gargs = types.SimpleNamespace(dataset=REF_DATASET[dataset], nz=nz, ngf=ngf, nc=nc, g_activation="lrelu",
                                  g_activation_leak=0.2, g_batchnorm=False)
    net = lsnf_amd._netG(gargs)
    for m in net.modules():
        if isinstance(m, nn.ConvTranspose2d):
            nn.init.xavier_normal_(m.weight)
    return net.tune() if tune else net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", choices=sorted(GEOMETRY), default="svhn")
    ap.add_argument("--batch", type=int, default=100, help="images per GPU (reference: 100, train.py:46)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="eager Langevin loop instead of the HIP-graph replay of one step")
    ap.add_argument("--no-tune", action="store_true", help="generator without channels-last / MIOpen find mode")
    ap.add_argument("--phases", action="store_true", help="diagnostic: synchronise after every phase of every iteration and print "
                    "the per-phase wall time of each (rank 0, stderr); the JSON line then carries their medians")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL, one GPU per rank); gloo lets several "
                    "ranks SHARE one GPU -- a rehearsal of the data-parallel code path on a one-GPU box, not a measurement")
    args = ap.parse_args()
    size, nz, ngf, f_width, K = GEOMETRY[args.dataset]
    step_size, sigma, nc, B = 0.1, 0.3, 3, args.batch

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = parallel.pick_device()                                # GPU of LOCAL_RANK (replaces get_free_gpu / nvidia-smi, train.py:708-714)
    torch.cuda.set_device(dev)
    rank, world, _ = parallel.init_from_env(dev, backend=args.backend)
    torch.manual_seed(1); np.random.seed(1)                     # same initial weights on every rank
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=f_width, f_flow_coupling=1)
    netG = make_generator(args.dataset, nz, ngf, nc, not args.no_tune).to(dev)
    if not args.no_tune:
        netG = netG.to(memory_format=torch.channels_last)
    netF = lsnf_amd._netF(hps, nz=nz).to(dev)
    parallel.broadcast_parameters(netF._param_list())
    if world > 1:
        netG = nn.parallel.DistributedDataParallel(netG, device_ids=[dev.index])
    optG = torch.optim.Adam(netG.parameters(), lr=3e-4, betas=(0.5, 0.999))
    try:                                                        # one fused Adam kernel over the flow's 60 tensors
        optF = torch.optim.Adam(netF.parameters(), lr=1e-4, betas=(0.5, 0.999), fused=True)
    except (RuntimeError, TypeError):
        optF = torch.optim.Adam(netF.parameters(), lr=1e-4, betas=(0.5, 0.999))
    mse = nn.MSELoss(reduction="sum")
    gen = torch.Generator(device=dev).manual_seed(100 + rank)   # every rank owns different rows
    x = torch.tanh(torch.randn(B, nc, size, size, device=dev, generator=gen))

    it = [0]
    sampler = None
    if world > 1 and args.backend == "gloo" and not args.no_graph:
        # ranks that SHARE one card (the gloo rehearsal): back-to-back graph launches of two processes on one device take seconds per
        # iteration in this ROCm build (3.9-7.9 s vs 33.8 ms eager) -- the rehearsal runs the eager loop
        if rank == 0:
            print("[train_synthetic] gloo backend = ranks share a GPU: eager Langevin loop (graph replay from several processes on one "
                  "device is pathologically slow here)", file=sys.stderr)
        args.no_graph = True
    if not args.no_graph:       # one Langevin step (generator gradient + fused flow update) captured once, replayed K times
        sampler = langevin.GraphedLangevinSampler(netG.module if world > 1 else netG, netF, B, nz, x.shape,
                                                  g_l_step_size=step_size, g_llhd_sigma=sigma, seed=1234, row0=rank * B)

    phase_log = []

    def iteration():
        marks = []

        def mark(name):                                         # --phases: wall time of the phase that just ended
            if args.phases:
                torch.cuda.synchronize()
                marks.append((name, time.perf_counter()))
        mark("start")
        z0 = torch.randn(B, nz, 1, 1, device=dev, generator=gen)
        gmod = netG.module if world > 1 else netG               # Langevin needs d/dz only: no gradient sync
        # Langevin noise drawn inside the update kernel: one Philox stream for the whole job, keyed by the GLOBAL row
        # (row0 = rank*B) and the global step count, so the draws do not depend on how the rows are sharded
        noise = lsnf_amd.flow.PhiloxNoise(seed=1234, offset=it[0] * K, row0=rank * B)
        it[0] += 1
        if sampler is not None:
            zk, ggn, gfn, f = sampler.run(z0, x, K, offset=noise.offset)
        else:
            zk, ggn, gfn, f = langevin.sample_langevin_post_z_with_flow(z0, x, gmod, netF, g_l_steps=K, g_l_step_size=step_size,
                                                                        g_llhd_sigma=sigma, g_l_with_noise=True, philox=noise)
        mark("langevin_K_steps")
        optG.zero_grad()
        loss_g = mse(netG(zk), x) / B                           # train.py:391-393 (DDP averages the gradients)
        mark("generator_forward")
        loss_g.backward()
        mark("generator_backward_incl_ddp_allreduce")
        optG.step()
        mark("generator_adam")
        optF.zero_grad(set_to_none=True)                        # train.py:404-415, fused: loss and the 60 gradients in 5 launches,
        loss_f = netF.mle_grads(zk.view(B, nz), max_norm=100.0 if world == 1 else None,   # then the one-bucket all-reduce
                                reuse_buffers=True)
        mark("flow_mle_grads")
        if world > 1:
            parallel.allreduce_gradients(netF.parameters(), average=True)
            mark("flow_grad_allreduce")
            torch.nn.utils.clip_grad_norm_(netF.parameters(), 100.0)
        optF.step()
        mark("flow_clip_adam")
        if args.phases:
            row = {b[0]: (b[1] - a[1]) * 1e3 for a, b in zip(marks, marks[1:])}
            phase_log.append(row)
            if rank == 0:
                print(f"iteration {len(phase_log):3d}: " + "  ".join(f"{k} {v:9.2f} ms" for k, v in row.items()), file=sys.stderr, flush=True)
        return loss_g.detach(), loss_f.detach()

    def wall(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out

    wall(iteration, args.warmup)
    # MIOpen's find mode searches its solvers the first time a convolution shape is seen -- seconds per shape, and the first
    # generator BACKWARD brings new shapes (6.2 s in iteration 1 of the two-rank rehearsal, profiles/r03_ddp_svhn_phases.txt).  With a
    # cold find database, or two ranks searching at once, that can outlast a fixed warm-up (the 12 s/iteration line of
    # profiles/r02_ddp_rehearsal_gloo.jsonl): warm up until two consecutive iterations agree, then time.
    settle, prev = 0, None
    for _ in range(20):
        t_one, _ = wall(iteration, 1)
        settle += 1
        if prev is not None and t_one < 1.5 * prev and prev < 1.5 * t_one:
            break
        prev = t_one
    ms_iter, (lg, lf) = wall(iteration, args.iters)
    z2d = torch.randn(B, nz, device=dev); gg = torch.randn(B, nz, device=dev); nn_ = torch.randn(B, nz, device=dev)
    ms_flow, _ = wall(lambda: netF.langevin_step(z2d, gg, nn_, step_size), 200)
    gmod = netG.module if world > 1 else netG
    z4 = torch.randn(B, nz, 1, 1, device=dev)
    ms_gen, _ = wall(lambda: lsnf_amd.netg.langevin_grad_g(gmod, z4, x, sigma), 20)
    zk = torch.randn(B, nz, 1, 1, device=dev)
    ms_mle, _ = wall(lambda: langevin.flow_mle_step(netF, optF, zk, f_max_norm=100.0, fused=True), 20)
    if world > 1:
        t = torch.tensor([ms_iter], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms_iter = t.item()
    if rank == 0:
        print(json.dumps({"config": f"{args.dataset} {size}x{size} nz={nz} ngf={ngf} f_width={f_width} g_l_steps={K} "
                                    f"B={B}/GPU x {world} GPU(s), synthetic x",
                          "ms_per_iteration": ms_iter, "iterations_per_s": 1e3 / ms_iter, "images_per_s": world * B * 1e3 / ms_iter,
                          "generator": "reference-shaped _netG" + ("" if args.no_tune else ", channels-last + MIOpen find mode"),
                          "langevin_loop": "eager" if args.no_graph else "one step captured in a HIP graph, replayed K times",
                          "generator_langevin_grad_ms": ms_gen,
                          "flow_langevin_step_ms": ms_flow, "flow_mle_step_ms": ms_mle,
                          "flow_share_of_iteration": (K * ms_flow + ms_mle) / ms_iter,
                          "loss_g": lg.item(), "loss_f": lf.item(), "warmup_iterations": args.warmup + settle,
                          **({"phase_median_ms": {k: float(np.median([r[k] for r in phase_log[args.warmup:]])) for k in phase_log[0]},
                              "phases_note": "--phases synchronises after every phase: ms_per_iteration of this run is not a timing"}
                             if args.phases and phase_log else {})}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
