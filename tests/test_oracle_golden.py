"""CPU: pins oracle/flow_oracle.py against the golden vectors the reference's own
model.py produced (tests/golden/make_golden.py).  fp32 results must agree with the
reference's fp32 results to rounding noise; the fp64 oracle must agree with the fp64 run."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import flow_oracle as O

CASES = golden_names()


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


@pytest.mark.parametrize("name", CASES)
def test_forward_logprob_matches_reference(name):
    p, g = load_golden(name)
    z = torch.from_numpy(g["z"])
    z1, logdet, ll = O.flow_log_prob(p, z)
    assert _rel(ll.numpy(), g["ll"]) <= 2e-6
    assert _rel(logdet.numpy(), g["logdet"]) <= 2e-6
    assert np.max(np.abs(z1.numpy() - g["z1"])) <= 2e-5 * max(1.0, np.abs(g["z1"]).max())


@pytest.mark.parametrize("name", CASES)
def test_fp64_oracle_matches_reference_fp64(name):
    p, g = load_golden(name)
    p64 = O.to_dtype(p, torch.float64)
    z1, logdet, ll = O.flow_log_prob(p64, torch.from_numpy(g["z"]).double())
    assert _rel(ll.numpy(), g["ll_f64"]) <= 1e-12
    assert np.max(np.abs(z1.numpy() - g["z1_f64"])) <= 1e-11


@pytest.mark.parametrize("name", CASES)
def test_per_block_outputs(name):
    p, g = load_golden(name)
    z = torch.from_numpy(g["z"])
    ld = torch.zeros(z.shape[0])
    for i in range(int(g["meta_depth"])):
        z, ld = O.block_fwd(p, i, z, ld)
        assert np.max(np.abs(z.numpy() - g["block_z"][i])) <= 2e-5 * max(1.0, np.abs(g["block_z"][i]).max())
        assert _rel(ld.numpy(), g["block_logdet"][i]) <= 2e-6


@pytest.mark.parametrize("name", CASES)
def test_grad_z(name):
    p, g = load_golden(name)
    gz = O.grad_neg_sum_ll_wrt_z(p, torch.from_numpy(g["z"]))
    ref = g["grad_z"]
    assert np.linalg.norm(gz.numpy() - ref) / np.linalg.norm(ref) <= 2e-6


@pytest.mark.parametrize("name", [n for n in CASES if "trained_s3" not in n])
def test_grad_params(name):
    p, g = load_golden(name)
    grads = O.grad_neg_mean_ll_wrt_params(p, torch.from_numpy(g["z"]))
    ref_keys = sorted(k[5:] for k in g if k.startswith("grad/"))
    assert sorted(grads) == ref_keys
    none_keys = sorted(k[9:] for k in g if k.startswith("gradnone/"))
    assert all(not O.is_live_param(k) for k in none_keys) and len(none_keys) == 10
    for k in ref_keys:
        a, b = grads[k].numpy(), g["grad/" + k]
        assert np.linalg.norm(a - b) <= 5e-5 * max(np.linalg.norm(b), 1e-3), k


@pytest.mark.parametrize("name", CASES)
def test_reverse(name):
    p, g = load_golden(name)
    B = int(g["meta_B"])
    x, nobj = O.flow_reverse(p, torch.from_numpy(g["rev_in"]), torch.zeros(B))
    scale = max(1.0, np.abs(g["rev_out"]).max())
    assert np.max(np.abs(x.numpy() - g["rev_out"])) <= 5e-4 * scale
    assert _rel(nobj.numpy(), g["rev_negobj"]) <= 1e-5
    rt, _ = O.flow_reverse(p, torch.from_numpy(g["z1"]), torch.zeros(B))
    assert np.max(np.abs(rt.numpy() - g["z"])) <= 2e-3 * max(1.0, np.abs(g["z"]).max())


def test_langevin_trajectory():
    p, g = load_golden("langevin_nz100_w64_B16_K3")
    z = torch.from_numpy(g["z0"])
    s = float(g["step_size"])
    for k in range(g["traj"].shape[0]):
        z, f, gf = O.langevin_prior_step(p, z, torch.from_numpy(g["grad_g"][k]), s)
        assert abs(f.item() - g["f_log_lkhd"][k]) <= 1e-5 * abs(g["f_log_lkhd"][k])
        assert np.linalg.norm(gf.numpy() - g["grad_f"][k]) <= 1e-5 * np.linalg.norm(g["grad_f"][k])
        assert np.max(np.abs(z.numpy() - g["traj"][k])) <= 1e-5


def test_state_dict_key_contract():
    p, _ = load_golden("tiny_nz8_w4_B7")
    assert sorted(p) == sorted(O.state_dict_keys(5))
    assert len(O.state_dict_keys(5)) == 85


def test_init_params_shapes_and_roundtrip():
    p = O.init_params(10, 6, depth=3, seed=3)
    assert O.depth_of(p) == 3
    z = torch.randn(5, 10)
    z1, ld, ll = O.flow_log_prob(p, z)
    back, nobj = O.flow_reverse(p, z1, torch.zeros(5))
    assert torch.allclose(back, z, atol=1e-4)
    assert torch.allclose(nobj, ld, atol=1e-4)
    # log-det equals log|det J| from autograd on a tiny case (fp64)
    p64 = O.to_dtype(p, torch.float64)
    z0 = torch.randn(1, 10, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda t: O.flow_forward(p64, t, torch.zeros(1, dtype=torch.float64))[0], z0)
    J = J.reshape(10, 10)
    _, ld64 = O.flow_forward(p64, z0, torch.zeros(1, dtype=torch.float64))
    assert abs(torch.log(torch.abs(torch.det(J))).item() - ld64.item()) < 1e-10


def test_philox_oracle_known_answers_and_moments():
    """The noise oracle (oracle/philox_oracle.py) against the Random123 known-answer vectors of Philox4x32-10
    (kat_vectors of the Random123 distribution: counter, key -> output), plus the moments of the normals."""
    from oracle.philox_oracle import philox4x32_10, langevin_noise
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in philox4x32_10(*ctr, *key)) == want
    n = langevin_noise(2048, 100, seed=11, offset=5)
    assert n.shape == (2048, 100) and np.isfinite(n).all()
    assert abs(n.mean()) < 0.01 and abs(n.var() - 1.0) < 0.02
    assert abs(np.mean(n ** 3)) < 0.05 and abs(np.mean(n ** 4) - 3.0) < 0.15
    # a pure function of (seed, offset, global row, column): row windows and offsets compose
    assert np.array_equal(langevin_noise(48, 100, 11, 5, row0=2000), n[2000:])
    assert not np.array_equal(langevin_noise(8, 100, 11, 6), n[:8])
    assert not np.array_equal(langevin_noise(8, 100, 12, 5), n[:8])
