"""GPU: the stash-writing instantiation of the pipelined forward (lsnf_fwd3q_kernel<.., STASH>) against the phase-separated kernel's
stash (math mode BF16X3_PHASED): block outputs, stash words, outputs; then forward-with-stash, backward and Langevin-step times."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, lsnf_amd
F = lsnf_amd.flow
dev = torch.device("cuda:0")
def t_us(fn, n=200):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
F.set_small_batch_max(0)
for nz, width, depth in ((bench.NZ, bench.WIDTH, bench.DEPTH), (128, 40, 3), (104, 64, 2), (128, 64, 1)):
    w = bench.synth_weights(1) if (nz, width, depth) == (bench.NZ, bench.WIDTH, bench.DEPTH) else None
    if w is None:
        from oracle import flow_oracle as O
        plan = lsnf_amd.prepare(lsnf_amd.params_from_state_dict(O.init_params(nz, width, depth, seed=5), depth, dev), nz, width, depth)
    else:
        plan = lsnf_amd.prepare([t.to(dev) for t in w], nz, width, depth)
    for B in (20000, 32768, 40001, 65536, 65537 + 300):
        z = torch.randn(B, nz, generator=torch.Generator().manual_seed(B)).to(dev)
        res = {}
        for name, mode in (("phased", F.MATH_BF16X3_PHASED), ("pipelined", F.MATH_BF16X3)):
            F.set_math_mode(mode)
            act = F.new_act_saved(plan, B, dev); act.fill_(float("nan"))
            saved = torch.full((max(depth - 1, 0), B, nz), float("nan"), device=dev)
            outs = (torch.empty_like(z), torch.empty(B, device=dev), torch.empty(B, device=dev))
            lsnf_amd.forward(plan, z, out=outs, act_saved=act, z_saved_out=saved)
            torch.cuda.synchronize()
            res[name] = (outs[0].clone(), outs[1].clone(), outs[2].clone(), saved, act)
        a, b = res["phased"], res["pipelined"]
        ai, bi = a[4].view(torch.int32), b[4].view(torch.int32)
        nbad = (ai != bi).sum().item()
        print(f"nz={nz} w={width} d={depth} B={B:6d}: z abs {(a[0]-b[0]).abs().max().item():.1e} ll abs {(a[2]-b[2]).abs().max().item():.1e} "
              f"z_saved abs {((a[3]-b[3]).abs().max().item() if depth > 1 else 0.0):.1e} nan in z_saved {torch.isnan(b[3]).sum().item()} "
              f"stash words differing {nbad} of {ai.numel()} (nan-filled words left: {(bi == ai.new_tensor(0x7fc00000)).sum().item()} vs {(ai == ai.new_tensor(0x7fc00000)).sum().item()})", flush=True)
w = bench.synth_weights(1)
plan = lsnf_amd.prepare([t.to(dev) for t in w], bench.NZ, bench.WIDTH, bench.DEPTH)
for B in (32768, 65536, 131072):
    zd = torch.randn(B, bench.NZ, device=dev); gg = torch.randn(B, bench.NZ, device=dev); nn_ = torch.randn(B, bench.NZ, device=dev)
    act = F.new_act_saved(plan, B, dev)
    outs = (torch.empty_like(zd), torch.empty(B, device=dev), torch.empty(B, device=dev))
    saved = torch.empty(bench.DEPTH - 1, B, bench.NZ, device=dev)
    for name, mode in (("phased", F.MATH_BF16X3_PHASED), ("pipelined", F.MATH_BF16X3)):
        F.set_math_mode(mode)
        fw = lambda: lsnf_amd.forward(plan, zd, out=outs, act_saved=act, z_saved_out=saved)
        f0 = lambda: lsnf_amd.forward(plan, zd, out=outs)
        fw()
        bw = lambda: lsnf_amd.backward_z(plan, outs[0], saved, ll_scale=-1.0, act_saved=act)
        lv = lambda: F.langevin_step(plan, zd, gg, nn_, 0.1, reuse_buffers=True)
        print(f"B={B:6d} {name:9s}: forward {t_us(f0):6.1f} us  forward+stash {t_us(fw):6.1f} us   backward from the stash {t_us(bw):6.1f} us   Langevin step {t_us(lv):6.1f} us", flush=True)
