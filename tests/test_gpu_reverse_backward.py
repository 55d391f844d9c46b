"""GPU parity of the reverse (sampling) kernel and the backward-w.r.t.-z kernel, through the C ABI,
against the reference's golden vectors and the oracle's autograd."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu
KINK = 2e-6   # see oracle.flow_oracle.relu_margin


@pytest.fixture(scope="module")
def lsnf():
    import lsnf_amd
    lsnf_amd.load_library()
    return lsnf_amd


@pytest.fixture(params=["recompute", "act-stash"])
def stash(request):
    """The backward either recomputes the coupling MLP from z_saved or reads the forward's activation stash."""
    return request.param == "act-stash"


def _fwd(lsnf, plan, z, stash, **kw):
    act = lsnf.flow.new_act_saved(plan, z.shape[0], z.device) if stash else None
    if act is not None:
        act.fill_(float("nan"))      # every word the backward reads must have been written by the forward
    return (*lsnf.forward(plan, z, save_for_backward=True, act_saved=act, **kw), act)


def _plan(lsnf, p, g, dev):
    nz, w, d = int(g["meta_nz"]), int(g["meta_width"]), int(g["meta_depth"])
    return lsnf.prepare(lsnf.params_from_state_dict(p, d, dev), nz, w, d, int(g.get("meta_coupling", 1)))


def inverse_tolerance(g, floor):
    """Tolerance of an inverse-pass comparison, relative to the data's scale: `floor` (what a well-conditioned flow
    achieves: the fuzz sweep holds the reverse to 5e-5, tools/fuzz_parity.py), or 3x the REFERENCE's own fp32 round-trip
    error reverse(forward(z)) - z stored in the fixture, whichever is larger -- the inverse of an ill-conditioned stack
    (the trained-like fixtures: reference round trip 1.4e-3) amplifies fp32 rounding in any implementation."""
    zmax = max(1.0, float(np.abs(g["z"]).max()))
    own = float(np.abs(g["roundtrip"] - g["z"]).max()) / zmax
    return max(floor, 3.0 * own)


@pytest.mark.parametrize("name", golden_names())
def test_reverse_matches_reference_golden(lsnf, kernels, gpu_device, name):
    p, g = load_golden(name)
    plan = _plan(lsnf, p, g, gpu_device)
    x, obj = lsnf.reverse(plan, torch.from_numpy(g["rev_in"]).to(gpu_device))
    x, negobj = x.cpu().numpy(), -obj.cpu().numpy()          # reference returns -objective (model.py:498)
    scale = max(1.0, np.abs(g["rev_out"]).max())
    assert np.max(np.abs(x - g["rev_out"])) <= inverse_tolerance(g, 5e-5) * scale
    assert np.max(np.abs(negobj - g["rev_negobj"]) / np.maximum(np.abs(g["rev_negobj"]), 1.0)) <= 1e-5


@pytest.mark.parametrize("name", golden_names())
def test_roundtrip_forward_reverse(lsnf, kernels, gpu_device, name):
    """forward o reverse = identity and the two log-dets cancel (the flow's own self-check)."""
    p, g = load_golden(name)
    plan = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    z1, ld, _, _ = lsnf.forward(plan, z)
    back, obj = lsnf.reverse(plan, z1, ld)
    zmax = max(1.0, z.abs().max().item())
    assert (back - z).abs().max().item() <= inverse_tolerance(g, 1e-4) * zmax
    assert (obj.abs() / ld.abs().clamp_min(1.0)).max().item() <= 2e-5


@pytest.mark.parametrize("name", golden_names())
def test_grad_z_matches_reference_golden(lsnf, kernels, stash, gpu_device, name):
    """d(-sum ll)/dz (train.py:320-323), fused ll_mode."""
    p, g = load_golden(name)
    plan = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    z1, ld, ll, saved, act = _fwd(lsnf, plan, z, stash)
    assert np.max(np.abs(ll.cpu().numpy() - g["ll"]) / np.maximum(np.abs(g["ll"]), 1.0)) <= 1e-5
    gz = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act).cpu().numpy()
    ref = g["grad_z"]
    # rows sitting on a ReLU kink (float64 pre-activation < 2e-6) have no well-defined fp32 gradient
    ok = (O.relu_margin(p, torch.from_numpy(g["z"])) > KINK).numpy()
    assert ok.sum() >= 0.97 * len(ok)
    assert np.linalg.norm(gz[ok] - ref[ok]) / np.linalg.norm(ref[ok]) <= 1e-5
    assert np.max(np.abs(gz[ok] - ref[ok])) <= 1e-4 * max(1.0, np.abs(ref).max())
    # a kink row may flip one ReLU: still the gradient of a neighbouring linear piece, hence close
    assert np.max(np.abs(gz - ref)) <= 2e-2 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nz,width,B", [(128, 64, 130), (100, 64, 77), (100, 128, 50), (20, 10, 33), (2, 1, 3)])
def test_backward_general_upstream_vs_oracle(lsnf, kernels, stash, gpu_device, nz, width, B):
    """Arbitrary upstream gradients (g_z1, g_logdet) against autograd over the oracle."""
    depth = 5
    p = O.init_params(nz, width, depth, seed=3 * nz + width)
    gen = torch.Generator().manual_seed(B)
    z = torch.randn(B, nz, generator=gen)
    gz1 = torch.randn(B, nz, generator=gen)
    gld = torch.randn(B, generator=gen)
    zz = z.clone().requires_grad_(True)
    z1r, ldr = O.flow_forward(p, zz, torch.zeros(B))
    (ref,) = torch.autograd.grad((z1r * gz1).sum() + (ldr * gld).sum(), zz)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z1, ld, _, saved, act = _fwd(lsnf, plan, z.to(gpu_device), stash, want_ll=False)
    ok = O.relu_margin(p, z) > KINK
    got = lsnf.backward_z(plan, z1, saved, gz1.to(gpu_device), gld.to(gpu_device), act_saved=act).cpu()
    assert ((got - ref)[ok].norm() / ref[ok].norm()).item() <= 1e-5
    # only one of the two upstream gradients
    z1r, ldr = O.flow_forward(p, zz, torch.zeros(B))
    (ref2,) = torch.autograd.grad((ldr * gld).sum(), zz)
    got2 = lsnf.backward_z(plan, z1, saved, None, gld.to(gpu_device), act_saved=act).cpu()
    assert ((got2 - ref2)[ok].norm() / ref2[ok].norm()).item() <= 1e-5


def test_langevin_trajectory_golden(lsnf, kernels, gpu_device):
    """Noise-free K-step Langevin trajectory captured from the reference (train.py:311-326) with the
    generator's gradient replayed from the fixture: pins the caller-side update around the flow."""
    p, g = load_golden("langevin_nz100_w64_B16_K3")
    plan = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z0"]).to(gpu_device)
    s = float(g["step_size"])
    for k in range(g["traj"].shape[0]):
        z1, ld, ll, saved = lsnf.forward(plan, z, save_for_backward=True)
        gf = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0)
        f = -ll.sum().item()
        assert abs(f - g["f_log_lkhd"][k]) <= 1e-5 * abs(g["f_log_lkhd"][k])
        assert np.linalg.norm(gf.cpu().numpy() - g["grad_f"][k]) <= 1e-5 * np.linalg.norm(g["grad_f"][k])
        z = z - 0.5 * s * s * (torch.from_numpy(g["grad_g"][k]).to(gpu_device) + gf)
        assert np.max(np.abs(z.cpu().numpy() - g["traj"][k])) <= 2e-5


def test_full_size_backward_properties(lsnf, gpu_device):
    """B=65536, nz=128: (a) linearity of the backward in the upstream gradient, (b) rows independent,
    (c) sampled rows vs the oracle."""
    nz, width, depth, B = 128, 64, 5, 65536
    p = O.init_params(nz, width, depth, seed=1)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(99))
    zd = z.to(gpu_device)
    z1, ld, ll, saved = lsnf.forward(plan, zd, save_for_backward=True)
    g = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0)
    g2 = lsnf.backward_z(plan, z1, saved, ll_scale=-2.0)
    assert (g2 - 2 * g).abs().max().item() <= 1e-4 * g.abs().max().item()
    ga = lsnf.backward_z(plan, z1, saved, z1.clone(), torch.full((B,), -1.0, device=gpu_device))
    assert torch.equal(ga, g)          # explicit upstream (z1, -1) == ll_mode(-1), bit for bit
    idx = torch.arange(5, B, 4099)
    ref = O.grad_neg_sum_ll_wrt_z(p, z[idx])
    ok = O.relu_margin(p, z[idx]) > KINK
    assert ((g.cpu()[idx] - ref)[ok].norm() / ref[ok].norm()).item() <= 1e-5
    # (d) the activation-stash backward agrees with the recomputing one (8-wave forward writes, 4-wave backward reads)
    z1b, _, llb, savedb, act = _fwd(lsnf, plan, zd, True)
    assert torch.equal(z1b, z1) and torch.equal(llb, ll)
    gs = lsnf.backward_z(plan, z1b, savedb, ll_scale=-1.0, act_saved=act)

    def same_up_to_kinks(a, b):
        # the stash holds the relu masks of the FORWARD's arithmetic (bf16x3 by default), the recomputing backward
        # takes them from its own fp32-MFMA recompute: a pre-activation within rounding of 0 may flip in a handful of
        # the 8.4e6 hidden units -- a neighbouring linear piece of the same function (oracle.relu_margin), not an error
        d = (a - b).abs().amax(1)
        gmax = b.abs().max().item()
        assert int((d > 2e-5 * gmax).sum()) <= 16 and d.max().item() <= 2e-2 * gmax
    same_up_to_kinks(gs, g)
    B2 = 40000 + 17                                  # ragged tail: last workgroup has idle waves
    z1c, _, _, savedc, actc = _fwd(lsnf, plan, zd[:B2].contiguous(), True)
    gc = lsnf.backward_z(plan, z1c, savedc, ll_scale=-1.0, act_saved=actc)
    same_up_to_kinks(gc, g[:B2])


@pytest.mark.parametrize("B", [5003, 10000, 16383])
def test_default_dispatch_mid_size_batch(lsnf, gpu_device, B):
    """Default dispatch between 4 096 and 16 384 rows: the bf16x3 latency forward in its 32- and 64-rows-per-workgroup forms
    (two / four 16-sample tiles per workgroup, ragged last workgroup: whole tiles past the batch) writes z_saved and the
    activation stash, the latency backward reads them -- what travels through HBM is layout-independent."""
    nz, width, depth = 100, 64, 5
    p = O.init_params(nz, width, depth, seed=5)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    assert lsnf.flow.set_math_mode(-1) == lsnf.flow.MATH_BF16X3 and lsnf.flow.set_small_batch_max(-1) == 16384
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(8))
    z1, ld, ll, saved, act = _fwd(lsnf, plan, z.to(gpu_device), True)
    g_stash = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act)
    g_rec = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0)
    idx = torch.arange(3, B, 101)
    ref = O.grad_neg_sum_ll_wrt_z(p, z[idx])
    ok = O.relu_margin(p, z[idx]) > KINK
    for g in (g_stash, g_rec):
        assert ((g.cpu()[idx] - ref)[ok].norm() / ref[ok].norm()).item() <= 1e-5
    _, _, llr = O.flow_log_prob(p, z[idx])
    assert ((ll.cpu()[idx] - llr).abs() / llr.abs()).max().item() <= 1e-5


def test_default_dispatch_large_batch(lsnf, gpu_device):
    """Default dispatch at 40 000 rows: the bf16x3 throughput forward (lsnf_fwd3.hip) writes z_saved and the activation
    stash, the bf16x3 throughput backward (lsnf_bwd3.hip) reads them; the Langevin step built from the two matches the
    oracle on rows away from a ReLU kink."""
    nz, width, depth, B = 128, 64, 5, 40000
    p = O.init_params(nz, width, depth, seed=6)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    assert lsnf.flow.set_math_mode(-1) == lsnf.flow.MATH_BF16X3
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(9))
    z1, ld, ll, saved, act = _fwd(lsnf, plan, z.to(gpu_device), True)
    g_stash = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act)
    idx = torch.arange(5, B, 211)
    ref = O.grad_neg_sum_ll_wrt_z(p, z[idx])
    ok = O.relu_margin(p, z[idx]) > KINK
    assert ((g_stash.cpu()[idx] - ref)[ok].norm() / ref[ok].norm()).item() <= 1e-5
    _, _, llr = O.flow_log_prob(p, z[idx])
    assert ((ll.cpu()[idx] - llr).abs() / llr.abs()).max().item() <= 1e-5
    zn, ll2, _, _ = lsnf.langevin_step(plan, z.to(gpu_device), None, None, 0.1)
    zr = z[idx].double() - 0.005 * ref.double()
    assert ((zn.cpu()[idx].double() - zr)[ok].abs().max() / zr.abs().max()).item() <= 2e-5


def test_fp16_split_range_guard_reverse(lsnf, gpu_device):
    """The throughput reverse on the two-term fp16 split (lsnf_rev2h.hip) under the same range guard as the forward: a
    row beyond fp16's range makes the bf16x3 pass behind it recompute that row's workgroup (bit-equal to LSNF_MATH_BF16X3
    there, the fp16 kernel's rows elsewhere), ordinary launches keep the fp16 kernel's results, and those match the oracle."""
    nz, width, depth, B = 128, 64, 5, 33000
    p = O.init_params(nz, width, depth, seed=31)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    g = torch.Generator().manual_seed(32)
    z = torch.randn(B, nz, generator=g)
    z_big = z.clone()
    z_big[777] *= 5.0e4
    obj = torch.randn(B, generator=g)
    prev_small = lsnf.flow.set_small_batch_max(0)
    prev = lsnf.flow.set_math_mode(lsnf.flow.MATH_BF16X3)
    try:
        ref_big = lsnf.reverse(plan, z_big.to(gpu_device), obj.to(gpu_device))
        ref = lsnf.reverse(plan, z.to(gpu_device), obj.to(gpu_device))
        lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2)
        got = lsnf.reverse(plan, z.to(gpu_device), obj.to(gpu_device))
        got_big = lsnf.reverse(plan, z_big.to(gpu_device), obj.to(gpu_device))
        got2 = lsnf.reverse(plan, z.to(gpu_device), obj.to(gpu_device))
        # (the huge row itself may legitimately overflow fp32 in the reverse direction -- z2 / sigmoid(p) with p << 0 --
        # so compare bit patterns: whatever LSNF_MATH_BF16X3 returns, the guarded fp16 mode returns the same)
        bits = lambda t: t.view(torch.int32)
        hot = slice(768, 1024)                                                   # the 256-row workgroup that holds row 777
        cold = torch.ones(B, dtype=torch.bool, device=gpu_device); cold[hot] = False
        assert torch.equal(bits(got_big[0][hot]), bits(ref_big[0][hot])) and torch.equal(bits(got_big[1][hot]), bits(ref_big[1][hot]))
        assert torch.equal(got_big[0][cold], got[0][cold]) and torch.equal(got_big[1][cold], got[1][cold])
        keep = torch.ones(B, dtype=torch.bool, device=gpu_device); keep[777] = False
        assert torch.isfinite(got_big[0][keep]).all() and torch.isfinite(got_big[1][keep]).all()
        assert torch.equal(got[0], got2[0]) and torch.equal(got[1], got2[1])
        assert not torch.equal(got[0], ref[0])                                   # the fp16 kernel did run
        assert (got[0] - ref[0]).abs().max().item() <= 2e-5 * ref[0].abs().max().item()
        idx = torch.arange(0, B, 97)
        xr, negobj = O.flow_reverse(O.to_dtype(p, torch.float64), z[idx].double(), obj[idx].double())
        assert (got[0].cpu()[idx].double() - xr).abs().max().item() <= 5e-5 * xr.abs().max().item()
        assert ((got[1].cpu()[idx].double() + negobj).abs() / negobj.abs().clamp_min(1.0)).max().item() <= 1e-5
    finally:
        lsnf.flow.set_math_mode(prev)
        lsnf.flow.set_small_batch_max(prev_small)


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
def test_dispatch_window_between_the_two_crossovers(lsnf, gpu_device, mode):
    """12 288 < B <= 16 384 with the built-in thresholds.  Every entry point uses ONE threshold (16 384 rows;
    12 288 in LSNF_MATH_FP16X2), so the family that writes z_saved / the activation stash is the family that reads them:
    forward + stash, backward from the stash, recomputing backward and the Langevin step at B = 14 000 against the oracle.
    Also: set(prev) restores the AUTO setting exactly (the old knob turned 'default' into 'set by the caller')."""
    nz, width, depth, B = 100, 64, 5, 14000
    p = O.init_params(nz, width, depth, seed=15)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    before = lsnf.flow.set_small_batch_max(lsnf.flow.SMALL_BATCH_AUTO)
    assert lsnf.flow.set_small_batch_max(4096) == lsnf.flow.SMALL_BATCH_AUTO       # the previous SETTING comes back, not a row count
    assert lsnf.flow.set_small_batch_max(lsnf.flow.SMALL_BATCH_AUTO) == 4096       # ... and AUTO again
    prev = lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2 if mode == "fp16x2" else lsnf.flow.MATH_BF16X3)
    try:
        assert lsnf.flow.set_small_batch_max(-1) == (12288 if mode == "fp16x2" else 16384)
        z = torch.randn(B, nz, generator=torch.Generator().manual_seed(16))
        z1, ld, ll, saved, act = _fwd(lsnf, plan, z.to(gpu_device), True)
        g_stash = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act)
        g_rec = lsnf.backward_z(plan, z1, saved, ll_scale=-1.0)
        idx = torch.arange(7, B, 131)
        ref = O.grad_neg_sum_ll_wrt_z(p, z[idx])
        ok = O.relu_margin(p, z[idx]) > KINK
        for g in (g_stash, g_rec):
            assert ((g.cpu()[idx] - ref)[ok].norm() / ref[ok].norm()).item() <= 1e-5
        _, _, llr = O.flow_log_prob(p, z[idx])
        assert ((ll.cpu()[idx] - llr).abs() / llr.abs()).max().item() <= 1e-5
        zn, ll2, _, _ = lsnf.langevin_step(plan, z.to(gpu_device), None, None, 0.1)
        zr = z[idx].double() - 0.005 * ref.double()
        assert ((zn.cpu()[idx].double() - zr)[ok].abs().max() / zr.abs().max()).item() <= 2e-5
        assert torch.equal(ll2, ll)
    finally:
        lsnf.flow.set_math_mode(prev)
        lsnf.flow.set_small_batch_max(before)
