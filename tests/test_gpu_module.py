"""GPU: `_netF` as a drop-in nn.Module -- forward / autograd w.r.t. z and parameters / reverse, all through the
C ABI, against the reference's golden vectors; plus the flow-MLE step (train.py:404-415) with Adam."""
import types

import os

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu
KINK = 2e-6


def make_net(lsnf, p, g, dev):
    nz, w, d = int(g["meta_nz"]), int(g["meta_width"]), int(g["meta_depth"])
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=d, f_flow_permutation=2, f_width=w,
                                f_flow_coupling=int(g.get("meta_coupling", 1)))
    net = lsnf._netF(hps, nz=nz)
    net.load_state_dict(p, strict=True)
    return net.to(dev), nz


@pytest.fixture(scope="module")
def lsnf():
    import lsnf_amd
    lsnf_amd.load_library()
    return lsnf_amd


def ll_of(z1, logdet):
    prior_ll = -0.5 * (z1 ** 2)                                  # train.py:317
    prior_ll = prior_ll.flatten(1).sum(-1) + np.log(2 * np.pi)   # train.py:318
    return prior_ll + logdet                                     # train.py:319


@pytest.mark.parametrize("name", golden_names())
def test_langevin_call_site(lsnf, kernels, gpu_device, name):
    """train.py:316-323 verbatim on the module: forward, log-prob in torch ops, autograd.grad w.r.t. z."""
    p, g = load_golden(name)
    net, nz = make_net(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device).view(-1, nz, 1, 1).clone().requires_grad_(True)
    z1, logdet, eps = net(torch.squeeze(z), objective=torch.zeros(int(z.shape[0])).to(gpu_device), init=False)
    assert eps == []
    ll = ll_of(z1, logdet)
    f_log_lkhd = -ll.sum()
    z_grad_f = torch.autograd.grad(f_log_lkhd, z)[0]
    assert z_grad_f.shape == z.shape
    assert np.max(np.abs(ll.detach().cpu().numpy() - g["ll"]) / np.abs(g["ll"])) <= 1e-5
    ok = (O.relu_margin(p, torch.from_numpy(g["z"])) > KINK).numpy()
    got = z_grad_f.view(-1, nz).cpu().numpy()
    assert np.linalg.norm(got[ok] - g["grad_z"][ok]) / np.linalg.norm(g["grad_z"][ok]) <= 1e-5


def _mle_backward(net, z2d, dev):
    """train.py:404-411 verbatim: loss_f = -ll.mean(); loss_f.backward()."""
    nz = z2d.shape[1]
    z_g_k = z2d.to(dev).view(-1, nz, 1, 1)
    net.zero_grad()
    z1, logdet, _ = net(torch.squeeze(z_g_k), objective=torch.zeros(int(z_g_k.shape[0])).to(dev), init=False)
    loss_f = -ll_of(z1, logdet).mean()
    loss_f.backward()
    return dict(net.named_parameters())


@pytest.mark.parametrize("name", [n for n in golden_names() if "trained_s3" not in n])
def test_flow_mle_call_site_param_grads(lsnf, kernels, gpu_device, name):
    """Parameter gradients of the flow-MLE step vs the reference's (golden) for all 60 live tensors.
    A row that sits on a ReLU kink (oracle.relu_margin) has no well-defined fp32 gradient: if the fixture
    holds such rows, the strict comparison runs on the kink-free rows against the oracle (itself pinned to
    the reference by tests/test_oracle_golden.py) and the full batch is compared loosely."""
    p, g = load_golden(name)
    net, nz = make_net(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"])
    ok = O.relu_margin(p, z) > KINK
    named = _mle_backward(net, z, gpu_device)
    strict = bool(ok.all())
    n_checked = 0
    for k in g:
        if k.startswith("gradnone/"):
            assert named[k[9:]].grad is None, k            # fc_1.b / fc_2.b stay untouched
        if not k.startswith("grad/"):
            continue
        ref = g[k]
        got = named[k[5:]].grad
        assert got is not None, k
        got = got.cpu().numpy()
        assert got.shape == ref.shape
        tol = (1e-4 if strict else 2e-2) * max(np.linalg.norm(ref), 1e-3)
        assert np.linalg.norm(got - ref) <= tol, (k, np.linalg.norm(got - ref), np.linalg.norm(ref))
        n_checked += 1
    assert n_checked == 60
    if not strict:
        assert ok.sum() >= 0.97 * len(ok)
        zs = z[ok]
        ref_grads = O.grad_neg_mean_ll_wrt_params(p, zs)
        named = _mle_backward(net, zs, gpu_device)
        for k, ref in ref_grads.items():
            got = named[k].grad.cpu()
            assert (got - ref).norm().item() <= 1e-4 * max(ref.norm().item(), 1e-3), k


@pytest.mark.parametrize("name", golden_names())
def test_reverse_call_sites(lsnf, kernels, gpu_device, name):
    """train.py:433-434 (return z) and model.py:495-498 (return_obj=True -> (z, -objective))."""
    p, g = load_golden(name)
    net, nz = make_net(lsnf, p, g, gpu_device)
    B = int(g["meta_B"])
    with torch.no_grad():
        eps = torch.from_numpy(g["rev_in"]).to(gpu_device)
        x = net(eps, objective=torch.zeros(B).to(gpu_device), reverse=True, return_obj=False)
        x2, nobj = net(eps, objective=torch.zeros(B).to(gpu_device), reverse=True, return_obj=True)
    assert torch.equal(x, x2)
    assert torch.equal(eps, torch.from_numpy(g["rev_in"]).to(gpu_device))       # functional: input untouched
    scale = max(1.0, np.abs(g["rev_out"]).max())
    zmax = max(1.0, float(np.abs(g["z"]).max()))
    tol = max(5e-5, 3.0 * float(np.abs(g["roundtrip"] - g["z"]).max()) / zmax)   # test_gpu_reverse_backward.inverse_tolerance
    assert np.max(np.abs(x.cpu().numpy() - g["rev_out"])) <= tol * scale
    assert np.max(np.abs(nobj.cpu().numpy() - g["rev_negobj"]) / np.maximum(np.abs(g["rev_negobj"]), 1.0)) <= 1e-5


def test_plan_cache_follows_optimizer_steps(lsnf, gpu_device):
    """Prepared weights are re-derived after every in-place parameter update (Adam step, load_state_dict),
    and the result tracks the oracle evaluated with the updated parameters."""
    p, g = load_golden("tiny_nz8_w4_B37_trained")
    net, nz = make_net(lsnf, p, g, gpu_device)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.5, 0.999))
    z = torch.from_numpy(g["z"]).to(gpu_device)
    losses = []
    for it in range(4):
        opt.zero_grad()
        z1, logdet, _ = net(z, objective=torch.zeros(z.shape[0], device=gpu_device))
        loss = -ll_of(z1, logdet).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)     # train.py:414
        opt.step()
        losses.append(loss.item())
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        _, _, ll_ref = O.flow_log_prob(sd, torch.from_numpy(g["z"]))
        with torch.no_grad():
            z1n, ldn, _ = net(z, objective=torch.zeros(z.shape[0], device=gpu_device))
        assert ((ll_of(z1n, ldn).cpu() - ll_ref).abs() / ll_ref.abs()).max().item() <= 1e-5
    assert losses[-1] < losses[0]                                    # MLE actually descends
    for k, v in net.named_parameters():                              # dead params never move (no grad)
        if k.endswith("fc_1.b") or k.endswith("fc_2.b"):
            assert v.grad is None and torch.count_nonzero(v) == 0


def test_fused_helpers_match_module_path(lsnf, gpu_device):
    p, g = load_golden("c3_nz128_w64_B200")
    net, nz = make_net(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    ll, grad = net.log_prob_and_grad(z, scale=-1.0)
    zz = z.clone().requires_grad_(True)
    z1, ld, _ = net(zz, objective=torch.zeros(z.shape[0], device=gpu_device))
    l2 = ll_of(z1, ld)
    (g2,) = torch.autograd.grad(-l2.sum(), zz)
    assert (ll - l2.detach()).abs().max().item() <= 2e-4
    assert ((grad - g2).norm() / g2.norm()).item() <= 1e-6


def test_params_and_z_grads_in_one_backward(lsnf, kernels, gpu_device):
    p, g = load_golden("c1_nz100_w64_B256")
    net, nz = make_net(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device).clone().requires_grad_(True)
    z1, ld, _ = net(z, objective=torch.zeros(z.shape[0], device=gpu_device))
    (-ll_of(z1, ld).sum()).backward()
    ok = (O.relu_margin(p, torch.from_numpy(g["z"])) > KINK).numpy()
    got = z.grad.cpu().numpy()
    assert np.linalg.norm(got[ok] - g["grad_z"][ok]) / np.linalg.norm(g["grad_z"][ok]) <= 1e-5
    B = z.shape[0]
    named = dict(net.named_parameters())
    k = O.block_prefix(2) + "invertible_1x1_conv.w"
    ref = g["grad/" + k] * B                                          # fixture holds d(-mean ll); here d(-sum ll)
    assert np.linalg.norm(named[k].grad.cpu().numpy() - ref) <= 1e-4 * np.linalg.norm(ref)


def test_parameter_update_between_forward_and_backward_raises_on_device(lsnf, gpu_device):
    """forward -> optimizer step -> backward must raise (PyTorch autograd would) instead of re-preparing the
    plan in place and pairing new weights with the activations saved from the old ones."""
    p, g = load_golden("c1_nz100_w64_B256")
    net, nz = make_net(lsnf, p, g, gpu_device)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    z = torch.from_numpy(g["z"]).to(gpu_device).requires_grad_(True)
    z1, logdet, _ = net(z, torch.zeros(z.shape[0], device=gpu_device))
    (-ll_of(z1, logdet).mean()).backward()
    g_first = z.grad.clone()
    opt.step()
    z1, logdet, _ = net(z, torch.zeros(z.shape[0], device=gpu_device))
    opt.step()                                    # out of order: the weights the forward used are gone
    with pytest.raises(lsnf.LsnfError, match="modified between forward and backward"):
        (-ll_of(z1, logdet).mean()).backward()
    # .data writes are invisible to the version counter: invalidate_plan() makes the next forward see them
    with torch.no_grad():
        _, ld0, _ = net(z.detach(), torch.zeros(z.shape[0], device=gpu_device))
    net._param_list()[1].data.add_(0.25)          # block 0 actnorm.logs: logdet moves by 3 * 0.25 * nz (+ what changes downstream)
    net.invalidate_plan()
    with torch.no_grad():
        _, ld1, _ = net(z.detach(), torch.zeros(z.shape[0], device=gpu_device))
    assert ((ld1 - ld0) - 3 * 0.25 * nz).abs().max().item() < 1.0
    assert torch.isfinite(g_first).all()


def test_generator_mirror_on_device_matches_reference_golden(gpu_device):
    """SURVEY 8f rank 4 on the GPU: `lsnf_amd._netG` (stock PyTorch-ROCm / MIOpen, tuned with channels-last + find mode)
    and `netg.langevin_grad_g` (train.py:312-314) against the reference's `_netG` golden vectors, every dataset variant --
    the generator half of the Langevin step as it runs in examples/train_synthetic.py, not only its CPU mirror.
    Runs in a FRESH child process (tests/generator_mirror_worker.py): MIOpen's find mode on the 256x256 variant launches a
    solver that faults or not depending on what else the process has allocated on the card (found in round 3: the same test,
    same MIOpen calls, aborted inside `conv_transpose2d` after the flow tests of a library build with slightly larger
    workspaces and passed after those of the previous build; with `AMD_SERIALIZE_KERNEL=3` the abort stayed at that call, so
    the faulting kernel is launched there, not by this repo's code).  The generator is stock MIOpen and out of the hot path;
    what this test pins is the mirror's numerics."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "generator_mirror_worker.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "generator_mirror_ok 10" in p.stdout, p.stdout[-4000:]


def test_parameter_gradients_run_to_run_spread_full_size(lsnf, gpu_device):
    """`lsnf_backward_params` contracts over the batch with fp32 atomics above 1 024 rows (csrc/lsnf_params.hip): the
    order of the adds, hence the last bits, differs from run to run.  Bound it at the headline size (B = 65 536, nz = 128):
    five runs on the same inputs agree per tensor to 2e-6 of the tensor's norm -- two orders below the 1e-4 the gradients
    are held to against the reference (the parity tests at golden sizes pin the VALUES; this test pins the SPREAD)."""
    nz, width, depth, B = 128, 64, 5, 65536
    p = O.init_params(nz, width, depth, seed=1)
    params = lsnf.params_from_state_dict(p, depth, gpu_device)
    plan = lsnf.prepare(params, nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(77)).to(gpu_device)
    z1, _, _, saved = lsnf.forward(plan, z, want_ll=False, save_for_backward=True)
    runs = []
    for _ in range(5):
        grads = lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B)
        runs.append([g.clone() for g in grads])
    torch.cuda.synchronize()
    worst = 0.0
    for k in range(len(runs[0])):
        ref = runs[0][k]
        nrm = max(ref.norm().item(), 1e-12)
        assert torch.isfinite(ref).all()
        for r in runs[1:]:
            worst = max(worst, (r[k] - ref).norm().item() / nrm)
    print("run-to-run spread of backward_params at B=65536: worst per-tensor rel-L2", worst)
    assert worst <= 2e-6


@pytest.mark.parametrize("nz,width,B", [(128, 64, 100), (128, 64, 5000), (100, 64, 4500), (50, 33, 4100), (100, 128, 4200),
                                         (128, 64, 14000), (128, 64, 40000), (128, 64, 65536),
                                         (128, 64, 40001), (128, 48, 20000), (104, 64, 20000), (100, 64, 20000), (64, 32, 17000)])
def test_parameter_gradients_from_the_stash_match_the_recomputing_path(lsnf, gpu_device, nz, width, B):
    """`lsnf_backward_params` fast path (forward keeps the activation stash and writes h1 / h2 into the workspace; the
    backward runs from the stash on the bf16 matrix pipe: latency kernels at B <= 16 384, throughput kernels above) against
    the recomputing fp32-MFMA path on the same inputs -- all 60 tensors and dL/dz; and against the float64 oracle on a
    whole batch at the sizes where that is affordable (these also cover the LDS-staged batch contraction of lsnf_params.hip,
    used from 4 096 rows, in its three row-vector widths: nz/width/half multiples of 4, of 2, odd; from 12 288 rows the contraction on
    the bf16 matrix pipe, lsnf_params3.hip -- with the row-major dump of the latency family (14 000 rows), with the tiled dump of the
    throughput family (whole tiles: 40 000, 65 536; a ragged last tile whose dead rows must read as zeros: 40 001; f_width 48, nz 64),
    row-major above the threshold where the geometry does not tile (nz = 104: the two-source G operand split at column 52), and the
    fp32 kernel where the bf16 one does not cover the rows (nz = 100: half = 50 is not a whole number of 16-byte groups))."""
    depth = 5
    p = O.init_params(nz, width, depth, seed=3)
    params = lsnf.params_from_state_dict(p, depth, gpu_device)
    plan = lsnf.prepare(params, nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B)).to(gpu_device)
    assert lsnf.flow.params_fast_path()
    act = lsnf.flow.new_act_saved(plan, B, gpu_device); act.fill_(float("nan"))
    ws = lsnf.flow.new_params_workspace(plan, B, gpu_device); ws.fill_(float("nan"))
    z1, _, _, saved = lsnf.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
    fast, gz_fast = lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, want_grad_z=True, act_saved=act, workspace=ws)
    fast = [g.clone() for g in fast]; gz_fast = gz_fast.clone()
    z1b, _, _, savedb = lsnf.forward(plan, z, want_ll=False, save_for_backward=True)
    assert torch.equal(z1, z1b)
    slow, gz_slow = lsnf.backward_params(plan, params, z, z1b, savedb, ll_scale=-1.0 / B, want_grad_z=True)
    # (the stash holds the ReLU masks of the forward's bf16x3 arithmetic, the recomputing path takes them from its own fp32 recompute:
    #  among tens of thousands of rows one pre-activation may sit within rounding of its kink and flip -- one sample's contribution to
    #  the gradients of that block's MLP, ~1e-4 of the tensor, identical for every contraction kernel (measured with LSNF_TN_X3=0 too):
    #  the MLP tensors of at most two blocks are tolerated)
    kinked = []
    for k, (a, b) in enumerate(zip(fast, slow)):
        assert torch.isfinite(a).all()
        rel = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        assert rel <= 1e-3, (k, lsnf.flow.BLOCK_PARAM_KEYS[k % 12], rel)
        if rel > 2e-5:
            kinked.append((k, lsnf.flow.BLOCK_PARAM_KEYS[k % 12], rel))
    assert len({k // 12 for k, _, _ in kinked}) <= (2 if B > 10000 else 0), kinked
    ok = (O.relu_margin(p, z.cpu()) > KINK).to(gpu_device)
    assert (gz_fast - gz_slow)[ok].norm().item() <= 1e-5 * gz_slow[ok].norm().item()
    if B <= 6000:
        ref = O.grad_neg_mean_ll_wrt_params(O.to_dtype(p, torch.float64), z.cpu().double())
        keys = [O.block_prefix(i) + k for i in range(depth) for k in lsnf.flow.BLOCK_PARAM_KEYS]
        for k, g in zip(keys, fast):
            r = ref[k].reshape(g.shape)
            # (thousands of rows: some pre-activation sits within fp32 rounding of a ReLU kink, where the two sides' gradients
            #  differ -- measured worst case 1.2e-4 on one fc_1.w at B = 4 200, identical for both contraction kernels)
            assert (g.cpu().double() - r).norm().item() <= (1e-4 if B <= 1000 else 5e-4) * max(r.norm().item(), 1e-9), k
    # run-to-run spread of the fast path (fp32 atomics in the batch contraction), as bounded for the recomputing one
    again = lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, act_saved=act, workspace=ws)
    for a, b in zip(fast, again):
        assert (a - b).norm().item() <= 2e-6 * max(a.norm().item(), 1e-12)


@pytest.mark.parametrize("mode", ["BF16X3_PHASED", "FP16X2"])
def test_parameter_gradients_fast_path_large_batch_other_modes(lsnf, gpu_device, mode):
    """The large-batch fast path in the other bf16x3-family modes: the phase-separated forward (tiled h dump, tag 1) and the fp16x2
    forward (row-major h dump, tag 0) in front of the same backward (tiled g arrays) and bf16x3 contraction -- the contraction reads
    the tag the forward left in the workspace."""
    F = lsnf.flow
    nz, width, depth, B = 128, 64, 5, 20000
    p = O.init_params(nz, width, depth, seed=3)
    params = lsnf.params_from_state_dict(p, depth, gpu_device)
    plan = lsnf.prepare(params, nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B)).to(gpu_device)
    prev = F.set_math_mode(getattr(F, "MATH_" + mode))
    try:
        act = F.new_act_saved(plan, B, gpu_device); act.fill_(float("nan"))
        ws = F.new_params_workspace(plan, B, gpu_device); ws.fill_(float("nan"))
        z1, _, _, saved = lsnf.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
        fast = [g.clone() for g in lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, act_saved=act, workspace=ws)]
        slow = lsnf.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B)
        for k, (a, b) in enumerate(zip(fast, slow)):
            assert torch.isfinite(a).all()
            assert (a - b).norm().item() <= 1e-3 * max(b.norm().item(), 1e-12), (k, F.BLOCK_PARAM_KEYS[k % 12])
    finally:
        F.set_math_mode(prev)
