"""CPU: the drop-in boundary -- `_netF` module surface (constructor, state_dict contract, init parity,
loud failure without a GPU) and the C ABI (library loads, exports every symbol the header declares)."""
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import flow_oracle as O

import lsnf_amd


def hps(width=64, depth=5, levels=1, perm=2, coupling=1):
    return types.SimpleNamespace(f_n_levels=levels, f_depth=depth, f_flow_permutation=perm, f_width=width,
                                 f_flow_coupling=coupling)


def test_state_dict_contract_matches_reference():
    net = lsnf_amd._netF(hps(64), nz=100)
    sd = net.state_dict()
    assert list(sd.keys()) == O.state_dict_keys(5)          # same keys, same registration order
    assert len(list(net.named_parameters())) == 70           # 85 keys, 70 named parameters (aliases de-duplicated)
    p, _ = load_golden("c1_nz100_w64_B256")
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(p[k].shape), k
    net.load_state_dict(p, strict=True)                      # a reference-keyed checkpoint loads unchanged
    for k, v in net.state_dict().items():
        assert torch.equal(v, p[k])
    pre = O.block_prefix(0)
    assert net.state_dict()[pre + "actnorm.b"].data_ptr() == net.state_dict()[pre + "actnorm.bias"].data_ptr()


@pytest.mark.parametrize("name,nz,width,seed", [("c1_nz100_w64_B256", 100, 64, 1), ("c5_nz100_w128_B100", 100, 128, 3),
                                                ("tiny_nz8_w4_B7", 8, 4, 11)])
def test_init_is_bitwise_the_reference_init(name, nz, width, seed):
    """Same distributions drawn in the same RNG order as reference model.py:176,230-233,318-319,340-342: under the
    same seeds the parameters equal the reference's (fc_zeros excluded: the fixtures perturb those)."""
    p, _ = load_golden(name)
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = lsnf_amd._netF(hps(width), nz=nz)
    for k, v in net.state_dict().items():
        if ".fc_zeros." in k:
            assert torch.count_nonzero(v) == 0
        else:
            assert torch.equal(v, p[k]), k


def test_constructor_errors_mirror_reference():
    with pytest.raises(NotImplementedError):
        lsnf_amd._netF(hps(levels=2), nz=8)                   # model.py:467-470
    with pytest.raises(Exception):
        lsnf_amd._netF(hps(perm=0), nz=8)                     # model.py:378-379
    add = lsnf_amd._netF(hps(coupling=0), nz=8)             # additive coupling: fc_zeros has nz/2 outputs (model.py:385)
    assert add.state_dict()[O.block_prefix(0) + "f.fc_zeros.w"].shape == (64, 4)


def test_cpu_call_fails_loudly_no_fallback():
    net = lsnf_amd._netF(hps(4), nz=8)
    with pytest.raises(lsnf_amd.LsnfError):
        net(torch.zeros(3, 8), objective=torch.zeros(3))
    with pytest.raises(lsnf_amd.LsnfError):
        net(torch.zeros(3, 8), objective=torch.zeros(3), reverse=True)
    with pytest.raises(NotImplementedError):
        net(torch.zeros(3, 8), objective=torch.zeros(3), init=True)
    with pytest.raises(ValueError):
        net(torch.zeros(8), objective=torch.zeros(1))          # the B == 1 squeeze quirk (train.py:316)


def test_module_protocol_used_by_train_py():
    net = lsnf_amd._netF(hps(4), nz=8)
    net.apply(lambda m: None)                                  # train.py:272
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)          # train.py:295 (no duplicate-parameter complaint)
    assert len(opt.param_groups[0]["params"]) == 70
    net.train(); net.eval()
    assert "revnet2d_step_s" in repr(net)
    live = net._param_list()
    assert len(live) == 60 and all(isinstance(t, torch.nn.Parameter) for t in live)


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lsnf_flow.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lsnf_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = lsnf_amd.load_library()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lsnf_flow.h but not exported"
    assert sorted(lsnf_amd.exported_symbols()) == declared      # the ctypes table covers the whole header
    assert lib.lsnf_abi_version() == 5


def test_geometry_queries_need_no_gpu():
    lib = lsnf_amd.load_library()
    assert lib.lsnf_plan_floats(128, 64, 5, 1) > 5 * 32768
    assert lib.lsnf_plan_floats(130, 64, 5, 1) == 0 and lib.lsnf_plan_floats(7, 4, 5, 1) == 0
    assert lib.lsnf_plan_floats(128, 64, 17, 1) == 0
    assert lib.lsnf_plan_floats(128, 64, 5, 0) == lib.lsnf_plan_floats(128, 64, 5, 1) and lib.lsnf_plan_floats(128, 64, 5, 2) == 0
    assert lib.lsnf_backward_params_workspace_floats(128, 64, 5, 100) > 0
    # the per-sample arrays of the parameter-gradient workspace are sized in whole 32-sample tiles (the large-batch kernels write whole
    # tiles): every batch size of a tile needs the same workspace, and a tile more needs 32 rows x 7 arrays x 5 blocks more
    ws = lambda B: lib.lsnf_backward_params_workspace_floats(128, 64, 5, B)
    assert ws(97) == ws(100) == ws(128) and ws(129) > ws(128)
    assert ws(160) - ws(128) == 5 * 32 * (128 + 64 + 64 + 64 + 64 + 64 + 64)


def test_pick_device_without_gpu_is_loud(monkeypatch):
    """parallel.pick_device replaces get_free_gpu (train.py:708-714, nvidia-smi): LOCAL_RANK decides under
    torch.distributed.run; with no GPU it raises instead of handing back a CPU device."""
    import torch
    from lsnf_amd import parallel
    if torch.cuda.device_count() == 0:
        with pytest.raises(RuntimeError):
            parallel.pick_device()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setenv("LOCAL_RANK", "6")
    assert parallel.pick_device() == torch.device("cuda", 2)
    monkeypatch.delenv("LOCAL_RANK")
    assert parallel.pick_device(prefer_free=False) == torch.device("cuda", 0)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda i: ((10, 50, 30, 20)[i], 100))
    assert parallel.pick_device() == torch.device("cuda", 1)


def _synthetic_image(B, size):
    b, c, i, j = np.meshgrid(np.arange(B), np.arange(3), np.arange(size), np.arange(size), indexing="ij")
    return torch.from_numpy(np.tanh(np.sin(0.37 * i + 0.91 * j + 1.7 * c + 2.3 * b)).astype(np.float32))


def test_generator_mirror_matches_reference_golden():
    """`lsnf_amd.netg._netG` (stock PyTorch / MIOpen, table-driven) against the reference's `_netG`
    (model.py:48-157) for every dataset variant: identical state_dict keys and shapes, image and Langevin z-gradient
    (train.py:312-314) equal to fp32 rounding.  Fixture: tests/golden/netg_variants.npz (make_golden.py netg)."""
    from lsnf_amd import netg
    raw = np.load(os.path.join(ROOT, "tests", "golden", "netg_variants.npz"), allow_pickle=False)
    tags = sorted({k.split("/")[0] for k in raw.files})
    assert len(tags) == 5
    for tag in tags:
        ds, act, bn = tag.rsplit("_", 2)
        size, nz, ngf, B, sub = (int(v) for v in raw[f"{tag}/meta"])
        args = types.SimpleNamespace(dataset=ds, nz=nz, ngf=ngf, nc=3, g_activation=act, g_activation_leak=0.2,
                                     g_batchnorm=bn == "bn1")
        net = netg._netG(args).eval()
        sd = {k[len(tag) + 4:]: torch.from_numpy(raw[k]) for k in raw.files if k.startswith(tag + "/sd/")}
        assert sorted(net.state_dict()) == sorted(sd)
        net.load_state_dict(sd, strict=True)
        z = torch.from_numpy(raw[f"{tag}/z"])
        x = _synthetic_image(B, size)
        with torch.no_grad():
            x_hat = net(z)
        assert x_hat.shape == (B, 3, size, size)
        assert (x_hat[:, :, ::sub, ::sub] - torch.from_numpy(raw[f"{tag}/x_hat"])).abs().max().item() <= 1e-5
        zg, gl = netg.langevin_grad_g(net, z, x, 0.3)
        ref = torch.from_numpy(raw[f"{tag}/z_grad_g"])
        assert abs(gl.item() - float(raw[f"{tag}/g_log_lkhd"])) <= 1e-5 * abs(float(raw[f"{tag}/g_log_lkhd"]))
        assert (zg - ref).norm().item() <= 1e-4 * ref.norm().item()
    with pytest.raises(ValueError):
        netg._netG(types.SimpleNamespace(dataset="mnist", nz=8, ngf=2, nc=1, g_activation="lrelu", g_batchnorm=False))


def _mock_kernels(monkeypatch):
    """Stand-ins for the HIP entry points so that the autograd bridge's bookkeeping (not its numbers) runs on CPU."""
    from lsnf_amd import netf
    calls = {"prepare": 0}

    def fake_prepare(tensors, nz, width, depth, coupling, plan=None):
        calls["prepare"] += 1
        return types.SimpleNamespace(device=tensors[0].device, nz=nz, depth=depth)
    monkeypatch.setattr(netf._netF, "_require_gpu", staticmethod(lambda params: None))
    monkeypatch.setattr(netf.flow, "prepare", fake_prepare)
    monkeypatch.setattr(netf.flow, "new_act_saved", lambda plan, B, dev: torch.zeros(1))
    monkeypatch.setattr(netf.flow, "forward", lambda plan, z, obj, **kw: (z * 2.0, obj + 1.0, None, torch.zeros(1, *z.shape)))
    monkeypatch.setattr(netf.flow, "backward_z", lambda plan, z1, saved, g1, gl, act_saved=None: torch.ones_like(z1))
    monkeypatch.setattr(netf.flow, "backward_params",
                        lambda plan, params, *a, **k: [torch.zeros_like(p) for p in params])
    return calls


def test_parameter_update_between_forward_and_backward_raises(monkeypatch):
    """the stale-weights guard must look at the LIVE parameters.  forward -> optimizer step ->
    backward used to pass the check (the cached key was only refreshed inside _plan()) and then re-prepare the plan in
    place: new weights against old activations.  PyTorch autograd raises in the same situation."""
    calls = _mock_kernels(monkeypatch)
    net = lsnf_amd._netF(hps(4, depth=2), nz=8)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    z = torch.randn(3, 8, requires_grad=True)
    # (a) the normal order works: forward, backward (z and parameters), step
    z1, ld, _ = net(z, torch.zeros(3))
    (z1.sum() + ld.sum()).backward()
    assert calls["prepare"] == 1 and torch.equal(z.grad, torch.ones(3, 8))
    opt.step()
    # (b) a step between forward and backward: both autograd nodes refuse
    z1, ld, _ = net(z, torch.zeros(3))
    assert calls["prepare"] == 2                        # the step above was seen by the forward
    opt.step()
    with pytest.raises(lsnf_amd.LsnfError, match="modified between forward and backward"):
        (z1.sum() + ld.sum()).backward()
    assert calls["prepare"] == 2                        # and the plan was NOT re-prepared under the pending backward
    # (c) load_state_dict between forward and backward
    z1, ld, _ = net(z, torch.zeros(3))
    net.load_state_dict(net.state_dict())
    with pytest.raises(lsnf_amd.LsnfError, match="modified between forward and backward"):
        torch.autograd.grad(z1.sum(), z)
    # (d) writes through .data are invisible to the version counter: invalidate_plan() is the documented way
    net(z, torch.zeros(3))
    n = calls["prepare"]
    net._param_list()[0].data.mul_(2.0)
    net(z, torch.zeros(3))
    assert calls["prepare"] == n
    net.invalidate_plan()
    net(z, torch.zeros(3))
    assert calls["prepare"] == n + 1


def test_philox_generator_advances_across_sampler_calls():
    """a `PhiloxNoise` kept across training iterations must not repeat its draws: `step()` is functional,
    `advance()` moves the object, and the K-step sampler advances the caller's generator by K when it returns."""
    from lsnf_amd import flow, langevin
    g = flow.PhiloxNoise(seed=5, offset=10, row0=3)
    h = g.step(4)
    assert (g.offset, h.offset, h.seed, h.row0) == (10, 14, 5, 3)
    assert g.advance(2) is g and g.offset == 12

    class FakeF:                                       # stands in for _netF.langevin_step (no GPU here): records the offsets it is given
        def __init__(self): self.offsets = []
        def langevin_step(self, z2d, gg, noise, s, reuse_buffers=False):
            self.offsets.append(noise.offset)
            return z2d - 0.0 * gg, torch.zeros(z2d.shape[0]), torch.zeros(z2d.shape[0]), torch.zeros(z2d.shape[0])
    netG = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(6, 12))
    f = FakeF()
    ph = flow.PhiloxNoise(seed=1, offset=100)
    z0, x = torch.randn(4, 6, 1, 1), torch.randn(4, 12)
    for _ in range(2):
        langevin.sample_langevin_post_z_with_flow(z0, x, netG, f, g_l_steps=3, g_l_step_size=0.1, g_llhd_sigma=0.3, philox=ph)
    assert f.offsets == [100, 101, 102, 103, 104, 105] and ph.offset == 106
