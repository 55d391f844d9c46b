"""GPU: the fused Langevin update (lsnf_langevin_step) and the caller-side harness (train.py:307-335, 404-415)."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsnf():
    import lsnf_amd
    lsnf_amd.load_library()
    return lsnf_amd


def make_net(lsnf, p, nz, w, d, dev):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=d, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    net = lsnf._netF(hps, nz=nz)
    net.load_state_dict(p, strict=True)
    return net.to(dev)


def test_fused_step_replays_reference_trajectory(lsnf, kernels, gpu_device):
    """The noise-free K=3 trajectory captured from the reference (its own _netG gradients replayed)."""
    p, g = load_golden("langevin_nz100_w64_B16_K3")
    net = make_net(lsnf, p, 100, 64, 5, gpu_device)
    z = torch.from_numpy(g["z0"]).to(gpu_device)
    s = float(g["step_size"])
    for k in range(g["traj"].shape[0]):
        gg = torch.from_numpy(g["grad_g"][k]).to(gpu_device)
        z_new, ll, gf_norm, gg_norm = net.langevin_step(z, gg, None, s)
        assert abs(-ll.sum().item() - g["f_log_lkhd"][k]) <= 1e-5 * abs(g["f_log_lkhd"][k])
        assert np.max(np.abs(z_new.cpu().numpy() - g["traj"][k])) <= 2e-5
        ref_gf = np.linalg.norm(g["grad_f"][k], axis=1)
        assert np.max(np.abs(gf_norm.cpu().numpy() - ref_gf) / ref_gf) <= 1e-5
        ref_gg = np.linalg.norm(g["grad_g"][k], axis=1)
        assert np.max(np.abs(gg_norm.cpu().numpy() - ref_gg) / ref_gg) <= 1e-5
        z = z_new


@pytest.mark.parametrize("nz,width,B", [(128, 64, 200), (100, 128, 37), (20, 10, 5)])
def test_fused_step_with_noise_and_inplace(lsnf, kernels, gpu_device, nz, width, B):
    p = O.init_params(nz, width, 5, seed=21)
    net = make_net(lsnf, p, nz, width, 5, gpu_device)
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(B, nz, generator=gen)
    gg = torch.randn(B, nz, generator=gen)
    noise = torch.randn(B, nz, generator=gen)
    s = 0.1
    ref, f_ref, gf_ref = O.langevin_prior_step(p, z, gg, s, noise)
    ok = (O.relu_margin(p, z) > 2e-6)
    zd = z.to(gpu_device)
    z_new, ll, gf_norm, _ = net.langevin_step(zd, gg.to(gpu_device), noise.to(gpu_device), s)
    assert torch.equal(zd.cpu(), z)                                   # not in place by default
    assert (z_new.cpu() - ref)[ok].abs().max().item() <= 2e-5
    assert abs(-ll.sum().item() - f_ref.item()) <= 1e-5 * abs(f_ref.item())
    assert ((gf_norm.cpu() - gf_ref.norm(dim=1))[ok].abs() / gf_ref.norm(dim=1)[ok]).max().item() <= 1e-5
    z_ip, _, _, _ = net.langevin_step(zd, gg.to(gpu_device), noise.to(gpu_device), s, inplace=True)
    assert z_ip.data_ptr() == zd.data_ptr() and torch.equal(z_ip, z_new)
    # no generator gradient, no noise
    z3, _, _, gg3 = net.langevin_step(z.to(gpu_device), None, None, s)
    ref3, _, _ = O.langevin_prior_step(p, z, None, s)
    assert gg3 is None and (z3.cpu() - ref3)[ok].abs().max().item() <= 2e-5


class ToyG(nn.Module):
    """Stand-in generator for the harness test (the reference's _netG is out of scope): z -> 3x4x4 'image'."""
    def __init__(self, nz):
        super().__init__()
        self.net = nn.Sequential(nn.ConvTranspose2d(nz, 8, 4, 1, 0), nn.LeakyReLU(0.2), nn.Conv2d(8, 3, 3, 1, 1), nn.Tanh())

    def forward(self, z):
        return self.net(z)


def test_sampler_and_mle_step_harness_vs_oracle(lsnf, gpu_device):
    """train.py:307-335 + 404-415 end to end (noise-free so that CPU and GPU can be compared)."""
    nz, w, B, K, s, sigma = 20, 12, 9, 4, 0.1, 0.3
    p = O.init_params(nz, w, 5, seed=8)
    torch.manual_seed(0)
    netG = ToyG(nz)
    x = torch.tanh(torch.randn(B, 3, 4, 4))
    z0 = torch.randn(B, nz, 1, 1)
    # CPU: oracle flow + the same generator
    mse = nn.MSELoss(reduction="sum")
    z = z0.clone()
    for _ in range(K):
        zz = z.clone().requires_grad_(True)
        g = 1.0 / (2.0 * sigma * sigma) * mse(netG(zz), x)
        gg = torch.autograd.grad(g, zz)[0].view(B, nz)
        znew, f_ref, gf = O.langevin_prior_step(p, z.view(B, nz), gg, s)
        z = znew.view(B, nz, 1, 1)
    # GPU: harness
    net = make_net(lsnf, p, nz, w, 5, gpu_device)
    zk, ggn, gfn, f = lsnf.langevin.sample_langevin_post_z_with_flow(
        z0.to(gpu_device), x.to(gpu_device), netG.to(gpu_device), net, g_l_steps=K, g_l_step_size=s,
        g_llhd_sigma=sigma, g_l_with_noise=False)
    assert zk.shape == (B, nz, 1, 1)
    assert (zk.cpu() - z).abs().max().item() <= 1e-4
    assert abs(f.item() - f_ref.item()) <= 1e-4 * abs(f_ref.item())
    assert abs(gfn.item() - gf.norm(dim=1).mean().item()) <= 1e-4 * gfn.item()
    # with noise: runs, changes z, stays finite
    zk2, _, _, _ = lsnf.langevin.sample_langevin_post_z_with_flow(
        z0.to(gpu_device), x.to(gpu_device), netG, net, g_l_steps=2, g_l_step_size=s, g_llhd_sigma=sigma)
    assert torch.isfinite(zk2).all() and not torch.equal(zk2, zk)
    # flow MLE step: the loss equals the oracle's and goes down over a few Adam steps
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, betas=(0.5, 0.999))
    _, _, ll_ref = O.flow_log_prob(p, z.view(B, nz))
    l0 = lsnf.langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0)
    assert abs(l0.item() - (-ll_ref.mean().item())) <= 1e-4 * abs(ll_ref.mean().item())
    for _ in range(5):
        l1 = lsnf.langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0)
    assert l1.item() < l0.item()
    # the fused step (netF.mle_grads) computes the same loss and gradients as the autograd restatement
    net.zero_grad(set_to_none=True)
    z2d = zk.reshape(B, nz)
    z1, ld, _ = net(z2d, objective=torch.zeros(B, device=gpu_device))
    loss_a = -(-0.5 * (z1 ** 2).sum(1) + float(np.log(2 * np.pi)) + ld).mean()
    loss_a.backward()
    ref = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
    net.zero_grad(set_to_none=True)
    loss_b = net.mle_grads(z2d)
    assert abs(loss_a.item() - loss_b.item()) <= 2e-6 * abs(loss_a.item())
    got = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
    assert set(got) == set(ref) and len(ref) == 60
    for k in ref:
        assert (got[k] - ref[k]).norm().item() <= 1e-5 * max(ref[k].norm().item(), 1e-3), k
    net.mle_grads(z2d, accumulate=True)
    for k in ref:
        assert (net.get_parameter(k).grad - 2 * ref[k]).norm().item() <= 2e-5 * max(ref[k].norm().item(), 1e-3), k
    # clipping on the flat buffer == torch's clip_grad_norm_ (train.py:413-414)
    total = float(torch.sqrt(sum((g ** 2).sum() for g in ref.values())))
    net.zero_grad(set_to_none=True)
    net.mle_grads(z2d, max_norm=0.5 * total)
    for k in ref:
        assert (net.get_parameter(k).grad - 0.5 * ref[k]).norm().item() <= 2e-5 * max(ref[k].norm().item(), 1e-3), k
    net.zero_grad(set_to_none=True)
    net.mle_grads(z2d, max_norm=10.0 * total)          # above the norm: untouched
    for k in ref:
        assert (net.get_parameter(k).grad - ref[k]).norm().item() <= 1e-5 * max(ref[k].norm().item(), 1e-3), k
    # reuse_buffers: the same gradient tensors every call, same values; a changed batch size re-allocates
    net.zero_grad(set_to_none=True)
    net.mle_grads(z2d, reuse_buffers=True)
    first = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
    for k in ref:
        first[k].add_(1.0)                                  # stale contents must be overwritten, not accumulated
    net.mle_grads(z2d, reuse_buffers=True)
    for k in ref:
        assert net.get_parameter(k).grad is first[k], k
        assert (first[k] - ref[k]).norm().item() <= 1e-5 * max(ref[k].norm().item(), 1e-3), k
    net.mle_grads(z2d[: B // 2], reuse_buffers=True)
    assert all(net.get_parameter(k).grad is not first[k] for k in ref)
    assert all(torch.isfinite(net.get_parameter(k).grad).all() for k in ref)
    l2 = lsnf.langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0, fused=True)
    l3 = lsnf.langevin.flow_mle_step(net, opt, zk, f_max_norm=100.0, fused=True)
    assert l3.item() < l2.item() <= l1.item()


@pytest.mark.parametrize("nz,width,B", [(100, 64, 77), (128, 64, 130), (20, 12, 33), (100, 128, 40)])
def test_in_kernel_philox_noise_matches_oracle(lsnf, kernels, gpu_device, nz, width, B):
    """lsnf_langevin_step with the in-kernel generator == the same step fed the oracle's noise tensor."""
    from oracle.philox_oracle import langevin_noise
    depth = 5
    p = O.init_params(nz, width, depth, seed=nz + width)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    gen = torch.Generator().manual_seed(B)
    z = torch.randn(B, nz, generator=gen).to(gpu_device)
    gg = torch.randn(B, nz, generator=gen).to(gpu_device)
    s, seed, off = 0.1, 0x1234_5678_9ABC_DEF1, (7 << 32) + 5
    rng = lsnf.flow.PhiloxNoise(seed, off)
    z_rng, ll_a, gf_a, gg_a = lsnf.langevin_step(plan, z, gg, rng, s)
    noise = torch.from_numpy(langevin_noise(B, nz, seed, off)).float().to(gpu_device)
    z_ten, ll_b, gf_b, gg_b = lsnf.langevin_step(plan, z, gg, noise, s)
    assert torch.equal(ll_a, ll_b) and torch.equal(gf_a, gf_b) and torch.equal(gg_a, gg_b)
    # v_log_f32 / v_sin_f32 / v_cos_f32 vs float64 libm: a few 1e-6 on a normal, times the step size
    assert (z_rng - z_ten).abs().max().item() <= 5e-6
    got = (z_rng - lsnf.langevin_step(plan, z, gg, None, s)[0]) / s          # the draws themselves
    assert (got - noise).abs().max().item() <= 1e-4
    # sharded rows with row0 draw what the full batch draws, bit for bit; other offsets / seeds do not
    cut = B // 3
    lo = lsnf.langevin_step(plan, z[:cut].contiguous(), gg[:cut].contiguous(), rng, s)[0]
    hi = lsnf.langevin_step(plan, z[cut:].contiguous(), gg[cut:].contiguous(),
                            lsnf.flow.PhiloxNoise(seed, off, row0=cut), s)[0]
    assert torch.equal(torch.cat([lo, hi]), z_rng)
    assert not torch.equal(lsnf.langevin_step(plan, z, gg, rng.step(), s)[0], z_rng)
    # device-side counter (for captured graphs): offset = host offset + *offset_dev
    ctr = torch.tensor([3], dtype=torch.int64, device=gpu_device)
    z_ctr = lsnf.langevin_step(plan, z, gg, lsnf.flow.PhiloxNoise(seed, off - 3, offset_dev=ctr), s)[0]
    assert torch.equal(z_ctr, z_rng)
    with pytest.raises(lsnf.LsnfError):
        lsnf.flow.langevin_step(plan, z, gg, lsnf.flow.PhiloxNoise(seed, off, row0=-1), s)


def test_in_kernel_philox_noise_full_size_moments(lsnf, gpu_device):
    """B = 65536 x nz = 128 draws (throughput kernel): moments, and no correlation between neighbours."""
    nz, width, depth, B = 128, 64, 5, 65536
    p = O.init_params(nz, width, depth, seed=1)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    base = lsnf.langevin_step(plan, z, None, None, 0.25)[0]
    n = ((lsnf.langevin_step(plan, z, None, lsnf.flow.PhiloxNoise(99, 1), 0.25)[0] - base) / 0.25).double()
    assert abs(n.mean().item()) < 2e-3 and abs(n.var().item() - 1.0) < 3e-3
    assert abs((n ** 4).mean().item() - 3.0) < 0.03
    assert abs((n[:, 1:] * n[:, :-1]).mean().item()) < 2e-3 and abs((n[1:] * n[:-1]).mean().item()) < 2e-3
    n2 = ((lsnf.langevin_step(plan, z, None, lsnf.flow.PhiloxNoise(99, 2), 0.25)[0] - base) / 0.25).double()
    assert abs((n * n2).mean().item()) < 2e-3


def test_graphed_sampler_matches_eager(lsnf, gpu_device):
    """One Langevin step captured in a HIP graph and replayed K times == the eager sampler, with the in-kernel noise
    advanced by the graph's own device counter; an optimizer step between two runs needs no re-capture."""
    nz, width, B, K, s, sigma = 20, 12, 37, 4, 0.1, 0.3
    p = O.init_params(nz, width, 5, seed=31)
    net = make_net(lsnf, p, nz, width, 5, gpu_device)
    torch.manual_seed(5)
    gargs = types.SimpleNamespace(dataset="svhn", nz=nz, ngf=4, nc=3, g_activation="lrelu", g_activation_leak=0.2,
                                  g_batchnorm=False)
    netG = lsnf._netG(gargs).to(gpu_device)
    gen = torch.Generator().manual_seed(6)
    z0 = torch.randn(B, nz, 1, 1, generator=gen).to(gpu_device)
    x = torch.tanh(torch.randn(B, 3, 32, 32, generator=gen)).to(gpu_device)
    sampler = lsnf.langevin.GraphedLangevinSampler(netG, net, B, nz, x.shape, g_l_step_size=s, g_llhd_sigma=sigma, seed=9)
    for trial in range(2):
        zg, ggn, gfn, f = sampler.run(z0, x, K, offset=100 * trial)
        ze, ggn_e, gfn_e, f_e = lsnf.langevin.sample_langevin_post_z_with_flow(
            z0, x, netG, net, g_l_steps=K, g_l_step_size=s, g_llhd_sigma=sigma,
            philox=lsnf.flow.PhiloxNoise(9, 100 * trial))
        assert (zg - ze).abs().max().item() <= 1e-5
        assert abs(f.item() - f_e.item()) <= 1e-5 * abs(f_e.item())
        assert abs(ggn.item() - ggn_e.item()) <= 1e-5 * ggn_e.item() and abs(gfn.item() - gfn_e.item()) <= 1e-5 * gfn_e.item()
        with torch.no_grad():                      # "optimizer step": both networks change in place
            for q in list(net.parameters()) + list(netG.parameters()):
                q.add_(0.01 * torch.randn_like(q))
