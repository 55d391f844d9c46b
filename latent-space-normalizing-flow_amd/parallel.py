"""Data-parallel evaluation of the flow prior: one process per GPU, rows of z sharded, weights replicated.

The per-sample flow evaluation needs no communication (every op of reference model.py:389-422 is
row-independent).  The only exchange is ONE all-reduce (sum) of a 3-float vector per evaluation:
[sum_b ll, sum_b logdet, row count] -- RCCL over xGMI on GPUs (`backend="nccl"` is RCCL on ROCm), gloo on
CPU in the tests.  For training steps the flow's parameter gradients (<= 0.94 MB) travel as ONE flat bucket.

The reference itself has no distributed code: this module adds what north_star asks for.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

STATS_DOUBLES = 264          # include/lsnf_flow.h LSNF_STATS_DOUBLES (= flow.STATS_DOUBLES; no import of the HIP binding here)


def init_from_env(device: Optional[torch.device] = None, backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world, local_rank).  No-op for WORLD_SIZE == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def pick_device(prefer_free: bool = True) -> torch.device:
    """Device for this process -- replaces the reference's `get_free_gpu()` (train.py:708-714), which shells out to
    `nvidia-smi`.  Under torch.distributed.run: the GPU of LOCAL_RANK (one process per GPU).  Standalone: the
    visible GPU with the most free memory (`torch.cuda.mem_get_info`, no external tool), or GPU 0 if
    `prefer_free` is False.  Raises if no GPU is visible: the flow has no CPU path."""
    n = torch.cuda.device_count()
    if n == 0:
        raise RuntimeError("no ROCm GPU visible: the flow prior has no CPU path")
    if "LOCAL_RANK" in os.environ:
        return torch.device("cuda", int(os.environ["LOCAL_RANK"]) % n)
    if not prefer_free or n == 1:
        return torch.device("cuda", 0)
    free = [torch.cuda.mem_get_info(i)[0] for i in range(n)]
    return torch.device("cuda", max(range(n), key=lambda i: free[i]))


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slab [start, stop) of `n_rows` owned by `rank`; slab sizes differ by at most one row."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_log_prob_stats(ll: torch.Tensor, logdet: torch.Tensor, group=None) -> torch.Tensor:
    """The single collective of the path.  Returns a float64 tensor [sum ll, sum logdet, rows] over ALL ranks.
    (float64 so that the reduced loss does not depend on the number of ranks beyond fp32 rounding of the
    per-rank partial sums.)"""
    stats = torch.stack([ll.sum(dtype=torch.float64), logdet.sum(dtype=torch.float64),
                         torch.tensor(float(ll.numel()), dtype=torch.float64, device=ll.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def reduce_sum_ll(ll: torch.Tensor, group=None) -> torch.Tensor:
    """Lean form of the collective for the hot loop (train.py:320 only needs sum ll): one on-device
    reduction + one all-reduce of a single float64.  Returns a 1-element float64 tensor."""
    s = ll.sum(dtype=torch.float64).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s


def reduce_stats_inplace(stats: torch.Tensor, group=None) -> torch.Tensor:
    """`stats` = the buffer filled by `flow.forward(..., stats=)`: all-reduces its public part
    [sum ll, sum logdet, rows] over the ranks (the single collective of the path) and returns that view.
    With one rank it is a no-op: the sums were already produced inside the forward kernel."""
    pub = stats[4:7]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pub, op=dist.ReduceOp.SUM, group=group)
    return pub


class PipelinedStatsReducer:
    """Hides the one collective of the path behind the following evaluations, and buckets it.

    The all-reduce of [sum ll, sum logdet, rows] is ~20 us of pure latency on 8 GPUs; nothing in the Langevin loop
    consumes the reduced value before the next flow evaluation starts (it is a logged diagnostic, train.py:320,332).
    So the sums of `bucket` consecutive evaluations are gathered in one bank (row j = the stats buffer of
    evaluation j, filled by the forward kernel itself) and travel in ONE asynchronous all-reduce per bank, on the
    communication stream, WHILE the next bank is being filled: every evaluation's sums are still reduced, but a
    collective kernel competes with the forward kernels for a CU (they fill the chip exactly: one workgroup per CU) only
    once per `bucket` evaluations.  Two banks alternate; a bank is handed out again only after the collective that
    reads it has been waited for (a stream-level wait for RCCL, no host sync).  bucket = 1: one collective per
    evaluation.

        red = PipelinedStatsReducer(device, bucket=8)
        for i in range(K):
            stats = red.next_buffer()               # row of the bank being filled
            flow.forward(plan, z, stats=stats)
            red.submit(stats)                       # every 8th call: async all-reduce of the bank
        totals = red.finish()                       # [sum ll, sum logdet, rows] of the last evaluation, reduced
    """

    def __init__(self, device, group=None, make_buffer=None, bucket: int = 1):
        if bucket < 1:
            raise ValueError("bucket must be >= 1")
        mk = make_buffer or (lambda n: torch.zeros(n, STATS_DOUBLES, dtype=torch.float64, device=device))
        self.bucket = bucket
        self.banks = [mk(bucket), mk(bucket)]
        for b in self.banks:             # the forward kernel WRITES whole rows (ABI v5: 264 doubles each): refuse anything else
            if tuple(b.shape) != (bucket, STATS_DOUBLES) or b.dtype != torch.float64 or not b.is_contiguous():
                raise ValueError(f"make_buffer(n) must return a contiguous float64 (n, {STATS_DOUBLES}) tensor, got "
                                 f"{tuple(b.shape)} {b.dtype}")
        # (row views made once: indexing a tensor costs ~2 us of host time, and a strong-scaling step has ~20 us of GPU work)
        self.rows = [[b[i] for i in range(bucket)] for b in self.banks]
        self.work = [None, None]
        self.pub = [None, None]      # reduced public parts of the banks (multi-rank runs)
        self.sent = [0, 0]           # rows of each bank that travelled in its collective
        self.bank = 0          # bank being filled
        self.fill = 0          # rows of it handed out and submitted
        self.group = group
        self.last = None       # (bank, row) of the last submitted evaluation

    def _multi(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def touches_stream(self) -> bool:
        """Will the next next_buffer() / submit() pair wait for or issue a collective (i.e. must it run with the evaluation's stream
        current)?  Only at the first and the last row of a bank: in between, a caller that launches on an explicit stream
        (`flow.BoundForward(stats, stream)`) can skip the `with torch.cuda.stream(...)` around the step."""
        return self.fill == 0 or self.fill == self.bucket - 1

    def next_buffer(self) -> torch.Tensor:
        if self.fill == 0:                                          # first row of a bank whose collective may be in flight
            self._complete(self.bank)
        return self.rows[self.bank][self.fill]

    def _complete(self, k: int) -> None:
        """Wait for bank k's collective (a stream-level wait for RCCL) and put the reduced sums back into its rows."""
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
            if self.pub[k] is not None:
                n = self.sent[k]
                self.banks[k][:n, 4:7].copy_(self.pub[k])   # only the rows that travelled: a row that was not submitted this
                self.pub[k] = None                            # round keeps its value (it was reduced once already)

    def _flush(self) -> None:
        if self.fill == 0:
            return
        if self._multi():
            # only the public part travels: [sum ll, sum logdet, rows] of every row of the bank, gathered into one contiguous
            # (bucket, 3) message (the rows are LSNF_STATS_DOUBLES wide since ABI v5: 2 KiB each, internal slots included);
            # of a partly filled bank (finish() only) just the submitted rows: the others hold sums that were reduced in an
            # earlier round, and reducing them again would multiply them by the world size
            self.sent[self.bank] = self.fill
            if self.fill == 1:
                # one evaluation per collective (bench.py's strong-scaling form): its three doubles are contiguous inside the row --
                # reduced in place, no gather / scatter copies (two launches and ~10 us of host time per evaluation less)
                self.pub[self.bank] = None
                self.work[self.bank] = dist.all_reduce(self.banks[self.bank][0, 4:7], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                self.pub[self.bank] = self.banks[self.bank][:self.fill, 4:7].contiguous()
                self.work[self.bank] = dist.all_reduce(self.pub[self.bank], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.bank ^= 1
        self.fill = 0

    def submit(self, stats: torch.Tensor) -> None:
        row = self.rows[self.bank][self.fill]
        assert stats is row or stats.data_ptr() == row.data_ptr(), "submit() must receive the buffer handed out by next_buffer()"
        self.last = (self.bank, self.fill)
        self.fill += 1
        if self.fill == self.bucket:
            self._flush()

    def finish(self) -> Optional[torch.Tensor]:
        self._flush()                                 # a partly filled bank travels too
        for k in (0, 1):
            self._complete(k)
        if self.last is None:
            return None
        b, r = self.last
        return self.banks[b][r][4:7]


def sharded_log_prob(evaluate: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                     z_local: torch.Tensor, group=None):
    """evaluate(z_local) -> (z1, logdet, ll) on this rank's rows (product: `_netF.log_prob`).
    Returns (z1, logdet, ll, stats) with stats = [global sum ll, global sum logdet, global rows]."""
    z1, logdet, ll = evaluate(z_local)
    return z1, logdet, ll, reduce_log_prob_stats(ll, logdet, group)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = True) -> int:
    """Sum (or average) the .grad of the given parameters over all ranks as ONE flat bucket (the flow has
    <= 233 760 fp32 parameters = 0.94 MB, below any sensible bucket size).  Parameters whose grad is None
    on every rank (fc_1.b, fc_2.b) are skipped.  Returns the number of elements reduced."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= dist.get_world_size(group)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return o


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None) -> None:
    """Replicate the flow weights from `src` to every rank (once, before the first step)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ps = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    with torch.no_grad():
        for p in ps:
            n = p.numel()
            p.copy_(flat[o:o + n].view_as(p))
            o += n
