"""CPU, world_size 2, gloo: the N > 1 path of the flow evaluation -- row sharding, the single all-reduce of
[sum ll, sum logdet, rows], one-bucket gradient all-reduce, parameter broadcast.  The per-shard evaluator
is the oracle here (tests may use it); on GPUs it is `_netF.log_prob` and the backend is RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import flow_oracle as O
import lsnf_amd
from lsnf_amd import parallel as P

NZ, WIDTH, DEPTH, B = 20, 12, 3, 37


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = P.init_from_env(torch.device("cpu"))
    assert (r, w) == (rank, world)
    p = O.init_params(NZ, WIDTH, DEPTH, seed=4)
    z = torch.randn(B, NZ, generator=torch.Generator().manual_seed(5))
    lo, hi = P.shard_bounds(B, world, rank)
    # (1) sharded log-prob + the one collective
    z1, ld, ll, stats = P.sharded_log_prob(lambda t: O.flow_log_prob(p, t), z[lo:hi])
    # (1b) the pipelined reducer (double-buffered async all-reduce) used by bench.py
    red = P.PipelinedStatsReducer(torch.device("cpu"))
    piped = []
    for it in range(5):
        st = red.next_buffer()
        if it >= 2:                                   # the buffer coming back holds step it-2, fully reduced
            piped.append(st[4:7].clone())
        st[4], st[5], st[6] = float(ll.sum()) * (it + 1), float(ld.sum()), float(ll.numel())
        red.submit(st)
    fin = red.finish().clone()
    # (1c) bucketed: the sums of 3 consecutive evaluations travel in one collective; 7 evaluations = 2 full banks + 1 row
    red3 = P.PipelinedStatsReducer(torch.device("cpu"), bucket=3)
    for it in range(7):
        st = red3.next_buffer()
        st[4], st[5], st[6] = float(ll.sum()) * (it + 1), float(ld.sum()), float(ll.numel())
        red3.submit(st)
    fin3 = red3.finish().clone()
    bank1 = red3.banks[1][:, 4:7].clone()            # evaluations 3, 4, 5, reduced
    # (1d) a partly filled bank over two finish() cycles: rows that were not submitted again must keep their once-reduced sums
    red2 = P.PipelinedStatsReducer(torch.device("cpu"), bucket=3)
    for it in range(4):                               # evaluations 0..2 fill bank 0, evaluation 3 sits alone in bank 1
        st = red2.next_buffer()
        st[4], st[5], st[6] = float(ll.sum()) * (it + 1), float(ld.sum()), float(ll.numel())
        red2.submit(st)
    red2.finish()
    st = red2.next_buffer()                           # second cycle: ONE evaluation, into row 0 of bank 0
    st[4], st[5], st[6] = float(ll.sum()) * 10, float(ld.sum()), float(ll.numel())
    red2.submit(st)
    fin2 = red2.finish().clone()
    bank0_after = red2.banks[0][:, 4:7].clone()      # row 0: evaluation "10" reduced; rows 1, 2: evaluations 1, 2, reduced ONCE
    # (2) broadcast: rank 1 starts from different weights and must end up with rank 0's
    q = O.init_params(NZ, WIDTH, DEPTH, seed=4 + rank)
    live = [q[k] for k in sorted(q) if O.is_live_param(k)]
    P.broadcast_parameters(live, src=0)
    same = all(torch.equal(q[k], p[k]) for k in sorted(q) if O.is_live_param(k))
    # (3) gradient bucket: local loss = -sum(ll_local)/B_global  -> summed grads == full-batch -mean(ll) grads
    keys = sorted(k for k in p if O.is_live_param(k))
    leaves = [torch.nn.Parameter(p[k].clone()) for k in keys]
    pp = dict(p); pp.update(dict(zip(keys, leaves)))
    _, _, ll_l = O.flow_log_prob(pp, z[lo:hi])
    (-ll_l.sum() / B).backward()
    dead = torch.nn.Parameter(torch.zeros(3))            # a parameter without grad must be skipped
    n = P.allreduce_gradients(leaves + [dead], average=False)
    if rank == 0:
        out.put({"stats": stats.tolist(), "same": same, "n": n, "piped": [p_.tolist() for p_ in piped], "fin": fin.tolist(), "fin3": fin3.tolist(), "bank1": bank1.tolist(), "fin2": fin2.tolist(), "bank0_after": bank0_after.tolist(), "ll": ll.tolist(), "lo_hi": (lo, hi),
                 "grads": {k: l.grad.numpy().copy() for k, l in zip(keys, leaves)}})
    else:
        out.put({"same": same, "lo_hi": (lo, hi)})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            b = [P.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_bounds(10, 2, 2)


def test_reducer_refuses_buffers_of_another_layout():
    """The forward kernel writes whole 264-double rows (ABI v5): a bank in the 8-double layout of ABI v4 would be overrun."""
    with pytest.raises(ValueError):
        P.PipelinedStatsReducer(torch.device("cpu"), make_buffer=lambda n: torch.zeros(n, 8, dtype=torch.float64))
    with pytest.raises(ValueError):
        P.PipelinedStatsReducer(torch.device("cpu"), make_buffer=lambda n: torch.zeros(n, P.STATS_DOUBLES, dtype=torch.float32))
    P.PipelinedStatsReducer(torch.device("cpu"), bucket=2)


def test_single_process_is_a_noop():
    ll = torch.arange(5.0)
    st = P.reduce_log_prob_stats(ll, 2 * ll)
    assert st.tolist() == [10.0, 20.0, 5.0]


def test_two_rank_gloo_matches_full_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [out.get(timeout=180) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    r0 = next(r for r in res if "stats" in r)
    assert all(r["same"] for r in res)
    p = O.init_params(NZ, WIDTH, DEPTH, seed=4)
    z = torch.randn(B, NZ, generator=torch.Generator().manual_seed(5))
    z1, ld, ll = O.flow_log_prob(p, z)
    assert abs(r0["stats"][0] - ll.double().sum().item()) <= 1e-6 * abs(ll.double().sum().item())
    assert abs(r0["stats"][1] - ld.double().sum().item()) <= 1e-6 * abs(ld.double().sum().item())
    assert r0["stats"][2] == B
    tot = ll.double().sum().item()
    for k, v in enumerate(r0["piped"]):                # steps 0,1,2 seen when their buffer is handed out again
        assert abs(v[0] - tot * (k + 1)) <= 1e-6 * abs(tot * (k + 1)) and v[2] == B
    assert abs(r0["fin"][0] - tot * 5) <= 1e-6 * abs(tot * 5) and r0["fin"][2] == B
    # bucketed reducer: evaluation 6 (alone in its bank) and evaluations 3..5 (one collective)
    assert abs(r0["fin3"][0] - tot * 7) <= 1e-6 * abs(tot * 7) and r0["fin3"][2] == B
    for j, row in enumerate(r0["bank1"]):
        assert abs(row[0] - tot * (4 + j)) <= 1e-6 * abs(tot * (4 + j)) and row[2] == B
    # partial final bank over two finish() cycles: the re-submitted row carries the new sums, the others were NOT reduced again
    assert abs(r0["fin2"][0] - tot * 10) <= 1e-6 * abs(tot * 10) and r0["fin2"][2] == B
    for j, mult in ((0, 10), (1, 2), (2, 3)):
        row = r0["bank0_after"][j]
        assert abs(row[0] - tot * mult) <= 1e-6 * abs(tot * mult) and row[2] == B, (j, row)
    lo, hi = r0["lo_hi"]
    assert torch.allclose(torch.tensor(r0["ll"]), ll[lo:hi], rtol=1e-6, atol=1e-5)
    ref = O.grad_neg_mean_ll_wrt_params(p, z)
    assert r0["n"] == sum(v.numel() for v in ref.values())
    for k, v in ref.items():
        assert (torch.from_numpy(r0["grads"][k]) - v).norm().item() <= 1e-5 * max(v.norm().item(), 1e-3), k


def test_reducer_says_when_it_touches_a_stream():
    """`touches_stream()`: only the first evaluation of a bank (the reducer may wait for the bank's previous collective) and the
    last one (it issues the collective) need the evaluation's stream to be current; bench.py names the stream in the launch itself
    for the ones in between."""
    import torch
    from lsnf_amd import parallel as P
    red = P.PipelinedStatsReducer(torch.device("cpu"), bucket=4,
                                  make_buffer=lambda n: torch.zeros(n, P.STATS_DOUBLES, dtype=torch.float64))
    seen = []
    for _ in range(9):
        seen.append(red.touches_stream())
        s = red.next_buffer()
        red.submit(s)
    assert seen == [True, False, False, True, True, False, False, True, True]
    one = P.PipelinedStatsReducer(torch.device("cpu"), bucket=1, make_buffer=lambda n: torch.zeros(n, P.STATS_DOUBLES, dtype=torch.float64))
    assert one.touches_stream()
