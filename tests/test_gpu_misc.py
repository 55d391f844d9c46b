"""GPU: depth sweep, hipGraph capture of the ABI calls, stream re-entrancy."""
import types

import numpy as np
import pytest
import torch

from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsnf():
    import lsnf_amd
    lsnf_amd.load_library()
    return lsnf_amd


@pytest.mark.parametrize("depth", [1, 2, 3, 8, 16])
def test_depth_sweep_forward_backward_reverse(lsnf, kernels, gpu_device, depth):
    """f_depth other than the default 5 (train.py:60), incl. the ABI maximum 16."""
    nz, w, B = 64, 48, 70
    p = O.init_params(nz, w, depth, seed=depth, fcz_std=0.03)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(depth))
    z1r, ldr, llr = O.flow_log_prob(p, z)
    gr = O.grad_neg_sum_ll_wrt_z(p, z)
    ok = O.relu_margin(p, z) > 2e-6
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, w, depth)
    z1, ld, ll, saved = lsnf.forward(plan, z.to(gpu_device), save_for_backward=True)
    assert (saved is None) == (depth == 1)
    assert ((ll.cpu() - llr).abs() / llr.abs().clamp_min(1.0)).max().item() <= 1e-5
    g = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0).cpu()
    assert ((g - gr)[ok].norm() / gr[ok].norm()).item() <= 1e-5
    back, obj = lsnf.reverse(plan, z1, ld)
    assert (back.cpu() - z).abs().max().item() <= 5e-3 and obj.abs().max().item() <= 1e-3 * max(1.0, ldr.abs().max().item())


def test_abi_calls_are_graph_capturable(lsnf, gpu_device):
    """The ABI promises no allocation / sync inside a call: a K-step noise-free Langevin loop (forward + fused
    update per step, reference train.py:311-326 without the generator) captured in one graph replays correctly."""
    nz, w, B, K, s = 100, 64, 100, 5, 0.1
    p = O.init_params(nz, w, 5, seed=31)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, 5, gpu_device), nz, w, 5)
    z0 = torch.randn(B, nz, generator=torch.Generator().manual_seed(1))
    ref = z0.clone()
    for _ in range(K):
        ref, _, _ = O.langevin_prior_step(p, ref, None, s)
    zbuf = z0.to(gpu_device).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on the capture stream (lazy module load etc.)
        lsnf.langevin_step(plan, zbuf.clone(), None, None, s)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        zc = zbuf
        for _ in range(K):
            zc, _, _, _ = lsnf.langevin_step(plan, zc, None, None, s)
        zout = zc
    zbuf.copy_(z0.to(gpu_device))
    graph.replay()
    torch.cuda.synchronize()
    ok = O.relu_margin(p, z0) > 1e-5
    assert (zout.cpu() - ref)[ok].abs().max().item() <= 5e-5
    z0b = torch.randn(B, nz, generator=torch.Generator().manual_seed(2))     # replay on new contents of the same buffer
    zbuf.copy_(z0b.to(gpu_device))
    graph.replay()
    torch.cuda.synchronize()
    refb = z0b.clone()
    for _ in range(K):
        refb, _, _ = O.langevin_prior_step(p, refb, None, s)
    assert (zout.cpu() - refb).abs().max().item() <= 1e-3


def test_two_streams_share_one_plan(lsnf, gpu_device):
    """Calls are re-entrant per stream: the same prepared weights serve two streams concurrently."""
    p = O.init_params(128, 64, 5, seed=5)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, 5, gpu_device), 128, 64, 5)
    torch.cuda.synchronize()
    zs = [torch.randn(3000 + 500 * i, 128, device=gpu_device) for i in range(2)]
    outs = [None, None]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for it in range(3):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = lsnf.forward(plan, zs[i])
    torch.cuda.synchronize()
    for i in range(2):
        ref = lsnf.forward(plan, zs[i])
        assert torch.equal(outs[i][0], ref[0]) and torch.equal(outs[i][2], ref[2])


def test_forward_refuses_undersized_or_foreign_buffers(lsnf, gpu_device):
    """The kernels write caller-owned buffers through bare pointers: `flow.forward` checks size, dtype and device of each
    (ABI v5 grew `stats` from 8 to 264 doubles and added the hidden-activation workspace: a buffer of the old layout must
    raise, not be overrun)."""
    F = lsnf.flow
    nz, w, d, B = 32, 16, 2, 40
    p = O.init_params(nz, w, d, seed=2)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, d, gpu_device), nz, w, d)
    z = torch.randn(B, nz, device=gpu_device)
    with pytest.raises(lsnf.LsnfError, match="stats"):
        lsnf.forward(plan, z, stats=torch.zeros(8, dtype=torch.float64, device=gpu_device))
    with pytest.raises(lsnf.LsnfError, match="stats"):
        lsnf.forward(plan, z, stats=torch.zeros(F.STATS_DOUBLES, dtype=torch.float32, device=gpu_device))
    with pytest.raises(lsnf.LsnfError, match="stats"):
        lsnf.forward(plan, z, stats=torch.zeros(F.STATS_DOUBLES, dtype=torch.float64))
    act = F.new_act_saved(plan, B, gpu_device)
    with pytest.raises(lsnf.LsnfError, match="act_saved"):
        lsnf.forward(plan, z, save_for_backward=True, act_saved=act[: act.numel() // 2])
    ws = F.new_params_workspace(plan, B, gpu_device)
    with pytest.raises(lsnf.LsnfError, match="params_ws"):
        lsnf.forward(plan, z, save_for_backward=True, act_saved=act, params_ws=ws[: ws.numel() - 8])
    with pytest.raises(lsnf.LsnfError, match="z_out"):
        lsnf.forward(plan, z, out=(torch.empty(B - 1, nz, device=gpu_device), torch.empty(B, device=gpu_device), torch.empty(B, device=gpu_device)))
    # and the well-formed call still runs
    st = F.new_stats(gpu_device)
    _, _, ll, _ = lsnf.forward(plan, z, save_for_backward=True, act_saved=act, params_ws=ws, stats=st)
    torch.cuda.synchronize()
    assert abs(st[4].item() - ll.double().sum().item()) <= 1e-9 * abs(ll.double().sum().item()) and st[6].item() == B


@pytest.mark.parametrize("B", [100, 6000])
def test_parameter_gradients_with_a_4_byte_aligned_z_in(lsnf, gpu_device, B):
    """The ABI promises 4-byte alignment only: z_in one float off a 16-byte boundary (the batch contraction of block 0 reads
    its rows) must give the gradients of the aligned call."""
    F = lsnf.flow
    nz, w, d = 128, 64, 3
    p = O.init_params(nz, w, d, seed=3)
    params = lsnf.params_from_state_dict(p, d, gpu_device)
    plan = lsnf.prepare(params, nz, w, d)
    z = torch.randn(B, nz, device=gpu_device)
    big = torch.empty(B * nz + 1, device=gpu_device)
    z_off = big[1:].view(B, nz)
    z_off.copy_(z)
    assert z_off.data_ptr() % 16 == 4 and z_off.is_contiguous()
    grads = []
    for zi in (z, z_off):
        z1, ld, ll, saved = lsnf.forward(plan, zi, save_for_backward=True)
        g = F.backward_params(plan, params, zi, z1, saved, ll_scale=-1.0 / B)
        grads.append([t.clone() for t in g])
    for a, b in zip(*grads):
        assert (a - b).norm().item() <= 2e-6 * max(a.norm().item(), 1e-6)


def test_bound_forward_is_the_same_launch(lsnf, gpu_device):
    """`flow.BoundForward` (buffers checked and pointers extracted once; bench.py's strong-scaling step): same outputs and in-kernel
    sums as `forward`, on whatever stream is current; a foreign `stats` buffer is refused."""
    F = lsnf.flow
    nz, width, depth, B = 128, 64, 5, 8192
    p = O.init_params(nz, width, depth, seed=2)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    ref_stats = F.new_stats(gpu_device)
    ref = lsnf.forward(plan, z, stats=ref_stats)
    out = (torch.empty_like(z), torch.empty(B, device=gpu_device), torch.empty(B, device=gpu_device))
    fw = F.BoundForward(plan, z, out)
    for o in out:
        o.fill_(float("nan"))
    st = F.new_stats(gpu_device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fw(st)
    side.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
    assert torch.equal(st[4:7], ref_stats[4:7])
    fw()                                                   # without sums
    torch.cuda.synchronize()
    assert torch.equal(out[2], ref[2])
    with pytest.raises(lsnf.LsnfError):
        fw(torch.zeros(8, dtype=torch.float64, device=gpu_device))
    with pytest.raises(lsnf.LsnfError):
        fw(torch.zeros(F.STATS_DOUBLES, dtype=torch.float32, device=gpu_device))
