"""GPU parity: HIP forward (through the C ABI) vs the golden vectors of the reference and vs the
oracle on the same seeded inputs.  Tolerances (north_star): log-prob <= 1e-5 relative, z1 <= 1e-4 abs."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu

LL_REL = 1e-5
Z_ABS = 1e-4


def _plan(lsnf, p, g, dev):
    nz, w, d = int(g["meta_nz"]), int(g["meta_width"]), int(g["meta_depth"])
    return lsnf.prepare(lsnf.params_from_state_dict(p, d, dev), nz, w, d, int(g.get("meta_coupling", 1))), nz, w, d


@pytest.fixture(scope="module")
def lsnf():
    import lsnf_amd
    lsnf_amd.load_library()
    return lsnf_amd


@pytest.mark.parametrize("name", golden_names())
def test_prepare_logdet_and_inverse(lsnf, gpu_device, name):
    p, g = load_golden(name)
    plan, nz, w, d = _plan(lsnf, p, g, gpu_device)
    lad = plan.logabsdet().cpu()
    winv = plan.winv().cpu()
    for i in range(d):
        W = p[O.block_prefix(i) + "invertible_1x1_conv.w"].double()
        ref = torch.linalg.slogdet(W)[1]
        assert abs(lad[i].item() - ref.item()) <= 1e-10 * max(1.0, abs(ref.item())) + 1e-11
        assert (winv[i] - torch.linalg.inv(W)).abs().max().item() <= 1e-9 * torch.linalg.inv(W).abs().max().item()


@pytest.mark.parametrize("name", golden_names())
def test_forward_matches_reference_golden(lsnf, kernels, gpu_device, name):
    p, g = load_golden(name)
    plan, nz, w, d = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    z1, ld, ll, _ = lsnf.forward(plan, z)
    torch.cuda.synchronize()
    z1, ld, ll = z1.cpu().numpy(), ld.cpu().numpy(), ll.cpu().numpy()
    scale = max(1.0, np.abs(g["z1"]).max())
    assert np.max(np.abs(ll - g["ll"]) / np.abs(g["ll"])) <= LL_REL
    assert np.max(np.abs(ld - g["logdet"]) / np.maximum(np.abs(g["logdet"]), 1.0)) <= LL_REL
    assert np.max(np.abs(z1 - g["z1"])) <= Z_ABS * scale
    # and not worse than ~10x the reference's own fp32 noise against its fp64 run
    err64 = np.max(np.abs(ll.astype(np.float64) - g["ll_f64"]) / np.abs(g["ll_f64"]))
    assert err64 <= 5e-6


@pytest.mark.parametrize("name", golden_names())
def test_per_block_launches_match(lsnf, kernels, gpu_device, name):
    """One launch per coupling block (first_block=i, n_blocks=1), chained through HBM."""
    p, g = load_golden(name)
    plan, nz, w, d = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    ld = torch.zeros(z.shape[0], device=gpu_device)
    for i in range(d):
        z, ld, _, _ = lsnf.forward(plan, z, ld, first_block=i, n_blocks=1, want_ll=False)
        ref = g["block_z"][i]
        assert np.max(np.abs(z.cpu().numpy() - ref)) <= Z_ABS * max(1.0, np.abs(ref).max())
        refl = g["block_logdet"][i]
        assert np.max(np.abs(ld.cpu().numpy() - refl) / np.maximum(np.abs(refl), 1.0)) <= LL_REL


def test_saved_activations_are_block_outputs(lsnf, kernels, gpu_device):
    p, g = load_golden("c3_nz128_w64_B200")
    plan, nz, w, d = _plan(lsnf, p, g, gpu_device)
    z = torch.from_numpy(g["z"]).to(gpu_device)
    _, _, _, saved = lsnf.forward(plan, z, save_for_backward=True)
    assert saved.shape == (d - 1, z.shape[0], nz)
    for i in range(d - 1):
        assert np.max(np.abs(saved[i].cpu().numpy() - g["block_z"][i])) <= Z_ABS * max(1.0, np.abs(g["block_z"][i]).max())


@pytest.mark.parametrize("nz,width,B", [(128, 64, 1), (128, 64, 33), (128, 64, 129), (100, 64, 300), (64, 32, 97),
                                        (20, 10, 130), (100, 128, 257), (2, 1, 5), (126, 127, 77)])
def test_forward_vs_oracle_ragged(lsnf, kernels, gpu_device, nz, width, B):
    """Seeded synthetic weights / inputs at ragged sizes (partial waves, partial workgroups, odd nz/2)."""
    depth = 5
    p = O.init_params(nz, width, depth, seed=nz + width)
    z = 1.5 * torch.randn(B, nz, generator=torch.Generator().manual_seed(B))
    obj = torch.randn(B, generator=torch.Generator().manual_seed(B + 1))
    z1r, ldr = O.flow_forward(p, z, obj)
    llr = O.log_prob(z1r, ldr)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z1, ld, ll, _ = lsnf.forward(plan, z.to(gpu_device), obj.to(gpu_device))
    assert ((ll.cpu() - llr).abs() / llr.abs().clamp_min(1.0)).max().item() <= LL_REL
    assert (z1.cpu() - z1r).abs().max().item() <= Z_ABS * max(1.0, z1r.abs().max().item())


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "bf16x3_phased", "fp16x2"])
def test_full_size_properties(lsnf, gpu_device, math):
    """BASELINE.json's full size (nz=128, w=64, B=65536), both arithmetic modes: size-independent properties.
    (a) row independence: the first 4096 rows of the big launch equal a 4096-row launch bit for bit;
    (b) a strided sample of rows equals the oracle within tolerance;
    (c) objective is additive: forward(z, obj) == forward(z, 0) + obj."""
    nz, width, depth, B = 128, 64, 5, 65536
    p = O.init_params(nz, width, depth, seed=1)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(1234))
    zd = z.to(gpu_device)
    prev = lsnf.flow.set_small_batch_max(8192)
    prev_math = lsnf.flow.set_math_mode({"fp32": lsnf.flow.MATH_FP32, "bf16x3": lsnf.flow.MATH_BF16X3,
                                         "bf16x3_phased": lsnf.flow.MATH_BF16X3_PHASED,
                                         "fp16x2": lsnf.flow.MATH_FP16X2}[math])
    z1, ld, ll, _ = lsnf.forward(plan, zd)                                   # throughput kernel
    z1s, lds, lls, _ = lsnf.forward(plan, zd[:16384].contiguous())           # throughput kernel, fewer rows
    assert torch.equal(z1[:16384], z1s) and torch.equal(ll[:16384], lls) and torch.equal(ld[:16384], lds)
    z1q, ldq, llq, _ = lsnf.forward(plan, zd[:4096].contiguous())            # latency kernel: same function, fp32 rounding apart
    assert (z1[:4096] - z1q).abs().max().item() <= 2e-5 and ((ll[:4096] - llq).abs() / llq.abs()).max().item() <= 2e-6
    z1p, ldp, llp, _ = lsnf.forward(plan, zd[:1024].contiguous())            # latency kernel is row independent bit for bit too
    assert torch.equal(z1q[:1024], z1p) and torch.equal(llq[:1024], llp)
    lsnf.flow.set_small_batch_max(prev)
    idx = torch.arange(0, B, 127)
    z1r, ldr, llr = O.flow_log_prob(p, z[idx])
    assert ((ll.cpu()[idx] - llr).abs() / llr.abs()).max().item() <= LL_REL
    assert (z1.cpu()[idx] - z1r).abs().max().item() <= Z_ABS
    obj = torch.full((B,), 3.25, device=gpu_device)
    _, ld2, _, _ = lsnf.forward(plan, zd, obj)
    assert (ld2 - (ld + 3.25)).abs().max().item() <= 2e-5
    assert torch.isfinite(ll).all()
    lsnf.flow.set_math_mode(prev_math)


@pytest.mark.parametrize("name", golden_names())
def test_split_bf16_is_fp32_faithful(lsnf, gpu_device, name):
    """The error-free three-way bf16 split (lsnf_fwd3.hip) is not a reduced-precision mode: against a float64
    evaluation of the oracle its log-prob error is of the size of the fp32-MFMA kernel's own (both ~3e-7 relative,
    the reference's fp32 noise floor), on every fixture incl. the trained-like ones."""
    p, g = load_golden(name)
    plan = _plan(lsnf, p, g, gpu_device)[0]
    z = torch.from_numpy(g["z"])
    _, _, ll64 = O.flow_log_prob(O.to_dtype(p, torch.float64), z.double())
    prev = lsnf.flow.set_small_batch_max(0)
    err = {}
    for mode, tag in ((lsnf.flow.MATH_FP32, "fp32"), (lsnf.flow.MATH_BF16X3, "bf16x3"), (lsnf.flow.MATH_BF16X3_PHASED, "bf16x3_phased"),
                      (lsnf.flow.MATH_FP16X2, "fp16x2")):
        prev_math = lsnf.flow.set_math_mode(mode)
        _, _, ll, _ = lsnf.forward(plan, z.to(gpu_device))
        lsnf.flow.set_math_mode(prev_math)
        err[tag] = ((ll.cpu().double() - ll64).abs() / ll64.abs().clamp_min(1.0)).max().item()
    lsnf.flow.set_small_batch_max(prev)
    print(name, err)
    assert max(err.values()) <= 2e-6, err
    assert max(err["bf16x3"], err["bf16x3_phased"]) <= 2.0 * err["fp32"] + 1e-7, err
    # the two-way fp16 split (lsnf_fwd2h.hip) drops terms of 2^-22 |w||x|: same class, slightly looser bound
    assert err["fp16x2"] <= 3.0 * err["fp32"] + 1e-7, err


def _wg_rows(row, B):
    """Rows of the workgroup (256 rows above 32 768, else 128) that holds `row`: the fix-up pass's unit of recomputation."""
    per = 256 if B > 128 * 256 else 128
    lo = (row // per) * per
    return slice(lo, min(lo + per, B))


def test_fp16_split_range_guard(lsnf, gpu_device):
    """LSNF_MATH_FP16X2 (opt-in) has fp16's exponent range.  A wave that meets an operand (or a folded weight) at or beyond
    65504 flags its first logdet element, and the bf16x3 pass queued behind the launch recomputes the workgroups that carry
    a flag: their rows are then bit-for-bit those of lsnf_fwd3b_kernel = LSNF_MATH_BF16X3_PHASED (no inf / NaN, no silently clipped ReLU input), every
    other row keeps the fp16 kernel's result.  The flag lives in the launch's own output, so launches of one plan in flight
    on several streams -- or replayed from several graphs -- cannot disturb each other."""
    nz, width, depth, B = 128, 64, 5, 33000
    p = O.init_params(nz, width, depth, seed=21)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    g = torch.Generator().manual_seed(22)
    z = torch.randn(B, nz, generator=g)
    z_big = z.clone()
    z_big[12345] *= 3.0e4                       # one row far outside fp16's range (|z| up to ~1e5)
    hot = _wg_rows(12345, B)
    cold = torch.ones(B, dtype=torch.bool, device=gpu_device); cold[hot] = False
    prev_small = lsnf.flow.set_small_batch_max(0)
    prev = lsnf.flow.set_math_mode(lsnf.flow.MATH_BF16X3_PHASED)

    def check_big(o, ref_big, h_small):
        assert torch.isfinite(o[2]).all()
        for k in range(3):
            assert torch.equal(o[k][hot], ref_big[k][hot])           # the flagged workgroup: recomputed by the bf16x3 kernel
            assert torch.equal(o[k][cold], h_small[k][cold])         # everybody else: the fp16 kernel's rows (row independence)

    try:
        stats = lsnf.flow.new_stats(gpu_device)
        ref_big = lsnf.forward(plan, z_big.to(gpu_device), stats=stats)
        ref_sum = stats[4].item()
        lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2)
        h_small = lsnf.forward(plan, z.to(gpu_device))
        got_big = lsnf.forward(plan, z_big.to(gpu_device))
        check_big(got_big, ref_big, h_small)
        assert not torch.equal(got_big[2][cold], ref_big[2][cold])            # (the fp16 kernel did run)
        # with in-kernel batch sums the call runs bf16x3 directly (a partial recomputation could not repair the sums)
        got_stats = lsnf.forward(plan, z_big.to(gpu_device), stats=stats)
        for a, b in zip(ref_big[:3], got_stats[:3]):
            assert torch.equal(a, b)
        assert stats[4].item() == ref_sum and stats[6].item() == B
        h_again = lsnf.forward(plan, z.to(gpu_device))                        # nothing sticks: the fp16 kernel's own results again
        for a, b in zip(h_small[:3], h_again[:3]):
            assert torch.equal(a, b)
        lsnf.flow.set_math_mode(lsnf.flow.MATH_BF16X3_PHASED)
        b_small = lsnf.forward(plan, z.to(gpu_device))
        assert not torch.equal(b_small[2], h_small[2])                        # (the two modes do differ in the last bits)
        assert ((b_small[2] - h_small[2]).abs() / b_small[2].abs().clamp_min(1.0)).max().item() <= 2e-6
        # launches of one plan in flight on two streams: the overflowing launch is repaired, its neighbours on the other
        # stream keep the fp16 kernel's results
        lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        zb, zs = z_big.to(gpu_device), z.to(gpu_device)
        torch.cuda.synchronize()
        outs_big, outs_small = [], []
        for _ in range(6):
            with torch.cuda.stream(s1):
                outs_big.append(lsnf.forward(plan, zb))
            with torch.cuda.stream(s2):
                outs_small.append(lsnf.forward(plan, zs))
        torch.cuda.synchronize()
        for o in outs_big:
            check_big(o, ref_big, h_small)
        for o in outs_small:
            assert torch.equal(o[2], h_small[2]) and torch.equal(o[0], h_small[0])
        # two CAPTURED launches of the same plan replayed at the same time on two streams, one of them overflowing
        # (with per-plan guard slots two graphs whose baked launch ids were congruent mod 127 shared a flag)
        bufs = [(torch.empty_like(zb), torch.empty(B, device=gpu_device), torch.empty(B, device=gpu_device)) for _ in range(2)]
        graphs = []
        for inp, out, st in ((zb, bufs[0], s1), (zs, bufs[1], s2)):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                lsnf.forward(plan, inp, out=out)                              # warm-up outside capture
                st.synchronize()
                with torch.cuda.graph(gr, stream=st):
                    lsnf.forward(plan, inp, out=out)
            graphs.append(gr)
        torch.cuda.synchronize()
        for _ in range(8):
            for out in bufs:
                for t in out:
                    t.fill_(float("nan"))
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                graphs[0].replay()
            with torch.cuda.stream(s2):
                graphs[1].replay()
            torch.cuda.synchronize()
            check_big(bufs[0], ref_big, h_small)
            for k in range(3):
                assert torch.equal(bufs[1][k], h_small[k])
        lsnf.flow.set_math_mode(lsnf.flow.MATH_BF16X3_PHASED)
        # an activation that leaves the range in mid-stack (block 0 scales by exp(6), |v| ~ 1e5 from |z| ~ 300) with
        # every weight and every input inside it: caught at the stage that consumes it
        q = dict(p)
        k = O.block_prefix(0) + "actnorm.logs"
        q[k] = torch.full_like(q[k], 2.0)
        plan_mid = lsnf.prepare(lsnf.params_from_state_dict(q, depth, gpu_device), nz, width, depth)
        z_mid = z.clone()
        z_mid[4321] *= 100.0
        ref_mid = lsnf.forward(plan_mid, z_mid.to(gpu_device))
        lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2)
        got_mid = lsnf.forward(plan_mid, z_mid.to(gpu_device))
        hot_mid = _wg_rows(4321, B)
        for a, b in zip(ref_mid[:3], got_mid[:3]):
            assert torch.equal(a[hot_mid].view(torch.int32), b[hot_mid].view(torch.int32))
        keep_rows = torch.ones(B, dtype=torch.bool, device=gpu_device); keep_rows[4321] = False
        assert torch.isfinite(got_mid[2][keep_rows]).all()
        assert ((got_mid[2] - ref_mid[2])[keep_rows].abs() / ref_mid[2][keep_rows].abs().clamp_min(1.0)).max().item() <= 1e-5   # (exp(6)-scaled block: |ll| ~ 3e6)
        lsnf.flow.set_math_mode(lsnf.flow.MATH_BF16X3_PHASED)
        # folded weights outside fp16's range (actnorm logs = 5 -> exp(15)): prepare marks the plan, every row is recomputed
        q = dict(p)
        k = O.block_prefix(2) + "actnorm.logs"
        q[k] = q[k].clone(); q[k][0, 7] = 5.0
        plan2 = lsnf.prepare(lsnf.params_from_state_dict(q, depth, gpu_device), nz, width, depth)
        ref2 = lsnf.forward(plan2, z.to(gpu_device))
        lsnf.flow.set_math_mode(lsnf.flow.MATH_FP16X2)
        got2 = lsnf.forward(plan2, z.to(gpu_device))
        assert torch.isfinite(got2[2]).all()
        for a, b in zip(ref2[:3], got2[:3]):
            assert torch.equal(a, b)
    finally:
        lsnf.flow.set_math_mode(prev)
        lsnf.flow.set_small_batch_max(prev_small)


def test_errors_are_loud(lsnf, gpu_device):
    p = O.init_params(8, 4, 5, seed=1)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, 5, gpu_device), 8, 4, 5)
    with pytest.raises(lsnf.LsnfError):
        lsnf.forward(plan, torch.zeros(4, 8))                      # CPU tensor
    with pytest.raises(lsnf.LsnfError):
        lsnf.forward(plan, torch.zeros(4, 10, device=gpu_device))  # wrong nz
    with pytest.raises(lsnf.LsnfError):
        lsnf.forward(plan, torch.zeros(4, 8, device=gpu_device), first_block=3, n_blocks=4)
    with pytest.raises(lsnf.LsnfError):
        lsnf.flow.alloc_plan(130, 64, 5, 1, gpu_device)            # geometry out of range
    z1, ld, ll, _ = lsnf.forward(plan, torch.zeros(0, 8, device=gpu_device))   # empty batch is legal
    assert z1.shape == (0, 8) and ll.shape == (0,)


@pytest.mark.parametrize("B", [4096, 4097, 9000, 16384, 16385, 24000, 32768, 32769, 40000])
def test_midsize_batches_default_dispatch(lsnf, gpu_device, B):
    """Untouched thresholds and default arithmetic: the stash-less forward crosses from the latency kernel to the pipelined
    throughput kernel at 4 096 rows and changes its workgroup shape at 16 384 and 32 768 rows (4 x 16, 8 x 16, 8 x 32 rows per
    workgroup) -- every shape against the oracle on a strided sample of rows, with and without in-kernel sums, and
    row-independent (a prefix launched alone on another shape gives the same rows to fp32 rounding)."""
    nz, width, depth = 128, 64, 5
    p = O.init_params(nz, width, depth, seed=5, fcz_std=0.05)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    assert lsnf.flow.set_small_batch_max(-1) == 16384 and lsnf.flow.set_math_mode(-1) == lsnf.flow.MATH_BF16X3
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B))
    zd = z.to(gpu_device)
    stats = lsnf.flow.new_stats(gpu_device)
    z1, ld, ll, _ = lsnf.forward(plan, zd)
    z1s, lds, lls, _ = lsnf.forward(plan, zd, stats=stats)
    assert torch.equal(z1, z1s) and torch.equal(ll, lls) and torch.equal(ld, lds)
    s = stats.cpu()
    assert abs(s[4].item() - ll.double().sum().item()) <= 1e-9 * abs(ll.double().sum().item()) and s[6].item() == B
    idx = torch.cat([torch.arange(0, B, 101), torch.tensor([B - 1])])
    z1r, ldr, llr = O.flow_log_prob(p, z[idx])
    assert ((ll.cpu()[idx] - llr).abs() / llr.abs().clamp_min(1.0)).max().item() <= LL_REL
    assert (z1.cpu()[idx] - z1r).abs().max().item() <= Z_ABS
    half = B // 2
    z1h, ldh, llh, _ = lsnf.forward(plan, zd[:half].contiguous())
    assert (z1[:half] - z1h).abs().max().item() <= 2e-5 and ((ll[:half] - llh).abs() / llh.abs().clamp_min(1.0)).max().item() <= 2e-6


@pytest.mark.parametrize("B", [1, 100, 129, 5000, 20000])
def test_in_kernel_batch_sums(lsnf, kernels, gpu_device, B):
    """stats: sum ll / sum logdet / rows accumulated by the kernel itself, re-armed for every launch (grids of more than 64
    workgroups go through the 64 sub-accumulators of the buffer: they too must be back at zero)."""
    p = O.init_params(128, 64, 5, seed=2)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, 5, gpu_device), 128, 64, 5)
    stats = lsnf.flow.new_stats(gpu_device)
    for rep in range(3):                       # repeated launches: the accumulator must come back to zero
        z = torch.randn(B, 128, device=gpu_device)
        z1, ld, ll, _ = lsnf.forward(plan, z, stats=stats)
        s = stats.cpu()
        assert abs(s[4].item() - ll.double().sum().item()) <= 1e-9 * abs(ll.double().sum().item()) + 1e-9
        assert abs(s[5].item() - ld.double().sum().item()) <= 1e-9 * abs(ld.double().sum().item()) + 1e-9
        assert s[6].item() == B and s[0].item() == 0.0 and s[1].item() == 0.0 and s[2].item() == 0.0
        assert s.numel() == lsnf.flow.STATS_DOUBLES and s[8:].view(torch.int64).abs().max().item() == 0
    lsnf.forward(plan, torch.zeros(0, 128, device=gpu_device), stats=stats)
    assert stats.cpu()[4:7].tolist() == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("z_scale,w_scale", [(1e-3, 1.0), (1.0, 1.0), (30.0, 1.0), (3.0, 4.0), (1.0, 0.05)])
def test_split_bf16_dynamic_range(lsnf, gpu_device, z_scale, w_scale):
    """bf16 keeps fp32's exponent range, so the error-free split has no range to manage: tiny / huge latents and
    re-scaled MLP weights (what training does to them) leave its error where the fp32-MFMA kernel's is, measured
    against a float64 evaluation of the oracle."""
    nz, width, depth, B = 128, 64, 5, 17000
    p = O.init_params(nz, width, depth, seed=11, fcz_std=0.05)
    g = torch.Generator().manual_seed(12)
    for k in list(p):
        if k.endswith("fc_1.w") or k.endswith("fc_2.w"):
            p[k] = p[k] * w_scale
        if k.endswith("actnorm.logs"):
            p[k] = p[k] + 0.1 * torch.randn(p[k].shape, generator=g)
    z = torch.randn(B, nz, generator=g) * z_scale
    idx = torch.arange(0, B, 61)
    _, _, ll64 = O.flow_log_prob(O.to_dtype(p, torch.float64), z[idx].double())
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    err = {}
    for mode, tag in ((lsnf.flow.MATH_FP32, "fp32"), (lsnf.flow.MATH_BF16X3, "bf16x3"), (lsnf.flow.MATH_BF16X3_PHASED, "bf16x3_phased"),
                      (lsnf.flow.MATH_FP16X2, "fp16x2")):
        prev = lsnf.flow.set_math_mode(mode)
        _, _, ll, _ = lsnf.forward(plan, z.to(gpu_device))
        lsnf.flow.set_math_mode(prev)
        assert torch.isfinite(ll).all()
        err[tag] = ((ll.cpu()[idx].double() - ll64).abs() / ll64.abs().clamp_min(1.0)).max().item()
    print(z_scale, w_scale, err)
    assert max(err.values()) <= 1e-5, err
    assert max(err["bf16x3"], err["bf16x3_phased"]) <= 2.0 * err["fp32"] + 1e-7, err
    assert err["fp16x2"] <= 3.0 * err["fp32"] + 1e-7, err


@pytest.mark.parametrize("nz,width,depth,B", [(128, 64, 5, 40001), (104, 48, 3, 33000), (128, 64, 1, 32800), (128, 64, 2, 70001),
                                               (40, 33, 2, 32800), (8, 64, 3, 20000), (32, 64, 4, 32769)])
def test_pipelined_forward_writes_the_phase_separated_stash(lsnf, gpu_device, nz, width, depth, B):
    """The stash-writing instantiation of the software-pipelined forward (lsnf_fwd3q_kernel<.., STASH>: buffer stores inside the
    MFMA phases, left in flight across the phase barriers) against the phase-separated kernel (math mode BF16X3_PHASED) on the
    same inputs: block outputs, sigma tiles and ReLU mask words bit for bit (what the backward reads does not depend on which
    forward wrote it), with every combination of the two optional buffers, ragged batches (waves and rows past the batch: their
    stores are dropped by the buffer descriptors), a padded second feature tile (nz = 104) and nz <= 64, where the first tile is
    padded too (the randomised sweep found the row stores of those geometries unguarded in this kernel's first version)."""
    F = lsnf.flow
    p = O.init_params(nz, width, depth, seed=11)
    plan = lsnf.prepare(lsnf.params_from_state_dict(p, depth, gpu_device), nz, width, depth)
    z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B)).to(gpu_device)
    prev_small, prev_math = F.set_small_batch_max(0), F.set_math_mode(-1)
    try:
        def run(mode, want_act, want_saved):
            F.set_math_mode(mode)
            act = F.new_act_saved(plan, B, gpu_device) if want_act else None
            if act is not None:
                act.fill_(float("nan"))
            saved = torch.full((depth - 1, B, nz), float("nan"), device=gpu_device) if want_saved else None
            guard = torch.full((4096,), 7.0, device=gpu_device)          # (allocated right behind: a store past the batch would land here)
            out = lsnf.forward(plan, z, act_saved=act, z_saved_out=saved)
            torch.cuda.synchronize()
            assert bool((guard == 7.0).all())
            return out[0], out[1], out[2], saved, act
        ref = run(F.MATH_BF16X3_PHASED, True, True)
        plain = run(F.MATH_BF16X3, False, False)
        for want_act, want_saved in ((True, True), (False, True), (True, False)):
            got = run(F.MATH_BF16X3, want_act, want_saved)
            assert torch.equal(got[0], ref[0])                              # z1: the same arithmetic in both kernels
            assert torch.equal(got[0], plain[0]) and torch.equal(got[1], plain[1]) and torch.equal(got[2], plain[2])   # stash or not: one kernel
            assert (got[2] - ref[2]).abs().max().item() <= 1e-4 * ref[2].abs().max().item()   # (the log-det sum is ordered differently)
            if want_saved and depth > 1:
                assert torch.equal(got[3], ref[3])
            if want_act:
                assert torch.equal(got[4].view(torch.int32), ref[4].view(torch.int32))
        # with the parameter-gradient h dump (STASH = 2 instantiation; tiled form from 12 288 rows): the whole workspace, bit for bit
        if B >= 12288 and depth > 1:
            def run_ws(mode):
                F.set_math_mode(mode)
                act = F.new_act_saved(plan, B, gpu_device); act.fill_(float("nan"))
                ws = F.new_params_workspace(plan, B, gpu_device); ws.fill_(float("nan"))
                saved = torch.full((depth - 1, B, nz), float("nan"), device=gpu_device)
                out = lsnf.forward(plan, z, act_saved=act, z_saved_out=saved, params_ws=ws)
                torch.cuda.synchronize()
                return out[0], saved, act, ws
            a, b = run_ws(F.MATH_BF16X3_PHASED), run_ws(F.MATH_BF16X3)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
            assert torch.equal(a[3].view(torch.int32), b[3].view(torch.int32))
    finally:
        F.set_small_batch_max(prev_small)
        F.set_math_mode(prev_math)
