"""Secondary measurements (SURVEY 8d 'secondary configs'): per-call latency of the flow path at the
reference's training batch size (B=100) and throughput of the other kernels at B=65536.  No oracle here."""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsnf_amd

dev = torch.device("cuda:0")

def timeit(fn, n=200, warm=20, chunks=5):
    """Median over `chunks` event-timed runs of n/chunks calls each (a single host hiccup -- GC, allocator, another
    tenant of the box -- otherwise lands in a 100-us average as +100 us)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    per = max(1, n // chunks)
    res = []
    for _ in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / per * 1e3)   # us
    return sorted(res)[len(res) // 2]

def make(nz, w, seed=1):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    torch.manual_seed(seed); np.random.seed(seed)
    net = lsnf_amd._netF(hps, nz=nz)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if ".fc_zeros." in n_: p_.add_(0.05 * torch.randn_like(p_))
    return net.to(dev)

out = []
for (tag, nz, w, B) in [("C2/C4 SVHN/CelebA nz=100 w=64 B=100", 100, 64, 100), ("C5 CelebA-HQ nz=100 w=128 B=100", 100, 128, 100),
                        ("C3 geometry nz=128 w=64 B=100", 128, 64, 100), ("C3 nz=128 w=64 B=65536", 128, 64, 65536),
                        ("C5 nz=100 w=128 B=65536", 100, 128, 65536)]:
    net = make(nz, w)
    z = torch.randn(B, nz, device=dev)
    gg = torch.randn(B, nz, device=dev)
    noise = torch.randn(B, nz, device=dev)
    obj = torch.zeros(B, device=dev)
    plan = net._plan()
    n = 200 if B <= 1000 else 30
    r = {"config": tag, "B": B}
    r["forward_logprob_us"] = timeit(lambda: lsnf_amd.forward(plan, z), n)
    if B > 16384:
        names = {0: "fp32", 1: "bf16x3", 2: "bf16x3_32", 3: "fp16x2", 4: "bf16x3_pipe"}
        pm = lsnf_amd.flow.set_math_mode(-1)
        r["default_math_mode"] = names[pm]
        for om in (0, 1, 3):
            if om != pm:
                lsnf_amd.flow.set_math_mode(om)
                r[f"forward_logprob_{names[om]}_us"] = timeit(lambda: lsnf_amd.forward(plan, z), n)
        lsnf_amd.flow.set_math_mode(pm)
    z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True)
    r["forward_saving_us"] = timeit(lambda: lsnf_amd.forward(plan, z, save_for_backward=True), n)
    r["backward_z_us"] = timeit(lambda: lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0), n)
    act = lsnf_amd.flow.new_act_saved(plan, B, dev)
    r["forward_saving_with_act_stash_us"] = timeit(lambda: lsnf_amd.forward(plan, z, save_for_backward=True, act_saved=act), n)
    r["backward_z_from_act_stash_us"] = timeit(lambda: lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0, act_saved=act), n)
    r["langevin_step_fwd_plus_fused_update_us"] = timeit(lambda: net.langevin_step(z, gg, noise, 0.1, reuse_buffers=True), n)
    r["reverse_us"] = timeit(lambda: lsnf_amd.reverse(plan, z), n)
    def mle():
        net.zero_grad(set_to_none=True)
        a, b, _ = net(z, objective=obj)
        (-(-0.5 * (a ** 2).sum(1) + 1.8378770664093453 + b).mean()).backward()
    r["mle_fwd_bwd_params_us"] = timeit(mle, max(10, n // 4), 5)
    def mle_fused():
        net.zero_grad(set_to_none=True)
        net.mle_grads(z)
    r["mle_fused_grads_us"] = timeit(mle_fused, max(10, n // 4), 5)
    def mle_fused_reuse():
        net.mle_grads(z, reuse_buffers=True)
    r["mle_fused_grads_reused_buffers_us"] = timeit(mle_fused_reuse, max(10, n // 4), 5)
    params = [p.detach() for p in net._param_list()]
    r["prepare_us"] = timeit(lambda: lsnf_amd.prepare(params, nz, w, 5, plan=plan), 20, 3)
    flop = 5 * (2 * nz * nz + 2 * (nz // 2 * w + w * w + w * nz)) * B
    r["forward_tflops"] = flop / r["forward_logprob_us"] / 1e6
    r["backward_z_tflops_counting_1.5x_fwd"] = 1.5 * flop / r["backward_z_us"] / 1e6
    out.append(r)
    print(json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/secondary.json", "w"), indent=1)
