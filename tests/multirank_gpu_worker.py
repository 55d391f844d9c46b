"""One rank of tests/test_gpu_multirank.py (a FRESH process: the pytest parent never hands it GPU state).  Both ranks use
cuda:0 and talk over gloo; each runs the product path on its contiguous slab of the batch (`parallel.shard_bounds`):
`_netF.log_prob` with in-kernel sums + `PipelinedStatsReducer(bucket=1)`, a sharded `langevin_step` with in-kernel Philox
noise keyed by the slab's first global row, `mle_grads` + `allreduce_gradients`.  Rank 0 also evaluates the whole batch alone
and writes the comparison to the JSON file named on the command line."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsnf_amd                                              # noqa: E402
from lsnf_amd import flow as F, parallel as P                # noqa: E402


def build_net(seed, dev, nz, width):
    torch.manual_seed(seed)
    np.random.seed(seed)
    hps = SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=width, f_flow_coupling=1)
    net = lsnf_amd._netF(hps, nz=nz)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "fc_zeros" in name:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return net.to(dev)


def main(out_path, B, nz, width):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    net = build_net(7 + rank, dev, nz, width)                # rank 1 starts from other weights ...
    P.broadcast_parameters(list(net.parameters()), src=0)    # ... and receives rank 0's
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(11)).to(dev)
    gg = 0.1 * torch.randn(B, nz, generator=torch.Generator().manual_seed(12)).to(dev)
    lo, hi = P.shard_bounds(B, world, rank)
    zl, ggl = z[lo:hi].contiguous(), gg[lo:hi].contiguous()

    # (1) sharded log-prob, sums produced inside the forward launch, one all-reduce per evaluation
    red = P.PipelinedStatsReducer(dev, bucket=1)
    st = red.next_buffer()
    z1l, ldl, lll = net.log_prob(zl, stats=st)
    red.submit(st)
    sums = red.finish().clone().cpu()
    # (2) sharded Langevin step: the in-kernel generator is keyed by the GLOBAL row
    znl, ll_in, gfl, ggn = net.langevin_step(zl, ggl, F.PhiloxNoise(seed=5, offset=3, row0=lo), step_size=0.1)
    # (3) flow-MLE gradients on the slab, scaled to the global batch, one flat bucket over the ranks
    net.zero_grad(set_to_none=True)
    net.mle_grads(zl)                                        # .grad = d(-mean over the SLAB of ll)/d theta
    with torch.no_grad():
        for p in net.parameters():
            if p.grad is not None:
                p.grad.mul_((hi - lo) / B)                   # -> this slab's share of d(-mean over the BATCH of ll)
    n_red = P.allreduce_gradients(net.parameters(), average=False)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()

    # gather the slabs on rank 0 (through the host: gloo)
    def gather(t):
        parts = [None] * world
        dist.all_gather_object(parts, t.cpu())
        return torch.cat(parts)
    z1_all, ll_all, zn_all, gf_all = gather(z1l), gather(lll), gather(znl), gather(gfl)
    if rank == 0:
        st1 = F.new_stats(dev)
        z1, ld, ll = net.log_prob(z, stats=st1)
        zn, _, gf, _ = net.langevin_step(z, gg, F.PhiloxNoise(seed=5, offset=3, row0=0), step_size=0.1)
        net.zero_grad(set_to_none=True)
        net.mle_grads(z)
        torch.cuda.synchronize()
        ref = {k: p.grad.detach() for k, p in net.named_parameters() if p.grad is not None}
        gerr = max(((grads[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-12)).item() for k in ref)
        res = {
            "world": world, "rows": [lo, hi], "n_reduced": n_red,
            "z1_bitwise": bool(torch.equal(z1_all, z1.cpu())), "ll_bitwise": bool(torch.equal(ll_all, ll.cpu())),
            "z_new_bitwise": bool(torch.equal(zn_all, zn.cpu())), "gf_norm_bitwise": bool(torch.equal(gf_all, gf.cpu())),
            "ll_max_rel": ((ll_all - ll.cpu()).abs() / ll.cpu().abs()).max().item(),
            "z_new_max_abs": (zn_all - zn.cpu()).abs().max().item(),
            "sum_ll": sums[0].item(), "sum_ll_single": st1[4].item(), "sum_ld": sums[1].item(), "sum_ld_single": st1[5].item(),
            "rows_total": sums[2].item(), "sum_ll_fp64_of_rows": ll.double().sum().item(),
            "grad_max_rel": gerr, "n_grad_tensors": len(ref),
        }
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
