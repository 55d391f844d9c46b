import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lsnf_amd
dev = torch.device("cuda:0")
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nz, w in ((100, 64), (100, 128), (128, 64)):
    hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=w, f_flow_coupling=1)
    net = lsnf_amd._netF(hps, nz=nz).to(dev)
    B = 100
    z = torch.randn(B, nz, device=dev); gg = torch.randn(B, nz, device=dev); noise = torch.randn(B, nz, device=dev)
    plan = net._plan()
    z1, ld, ll, saved = lsnf_amd.forward(plan, z, save_for_backward=True)
    t_b = timeit(lambda: lsnf_amd.backward_z(plan, z1, saved, ll_scale=-1.0))
    lib = lsnf_amd.load_library()
    import ctypes
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    zn = torch.empty_like(z); gf = torch.empty(B, device=dev); g2 = torch.empty(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def lv(noise_=noise, gg_=gg):
        rc = lib.lsnf_langevin_step(P(plan.buf), nz, w, 5, 1, B, P(z), P(z1), P(saved), P(gg_), P(noise_), 0.1, P(zn), P(gf), P(g2), st)
        assert rc == 0
    print(nz, w, "bwd", round(t_b, 1), "langevin tail only", round(timeit(lv), 1), "no noise/gg", round(timeit(lambda: lv(None, None)), 1),
          "full python step", round(timeit(lambda: net.langevin_step(z, gg, noise, 0.1)), 1))
