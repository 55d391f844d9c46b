"""Forward / backward time of the throughput kernels across geometries at B = 65536 (I/O path check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lsnf_amd
dev = torch.device("cuda:0")
def weights(nz, w, depth=5, seed=1):
    g = torch.Generator().manual_seed(seed); rs = np.random.RandomState(seed); half = nz // 2; out = []
    for _ in range(depth):
        rn = lambda *s: torch.randn(*s, generator=g) * 0.05
        q = torch.tensor(np.linalg.qr(rs.randn(nz, nz))[0], dtype=torch.float32)
        out += [rn(nz), rn(nz), q, rn(half, w), rn(w), rn(w), rn(w, w), rn(w), rn(w), rn(w, nz), rn(nz), rn(nz)]
    return [t.to(dev) for t in out]
def t(fn, n=400, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
B = 65536
for nz, w in ((128, 64), (100, 64), (96, 64), (100, 128), (64, 64), (126, 64)):
    plan = lsnf_amd.prepare(weights(nz, w), nz, w, 5)
    z = torch.randn(B, nz, device=dev)
    z1, ld, ll, sv = lsnf_amd.forward(plan, z, save_for_backward=True)
    flop = 5 * (2 * nz * nz + 2 * (nz // 2 * w + w * w + w * nz)) * B
    tf = t(lambda: lsnf_amd.forward(plan, z)); tb = t(lambda: lsnf_amd.backward_z(plan, z1, sv, ll_scale=-1.0), 200, 100)
    print(f"nz={nz:3d} w={w:3d} half%4={nz//2%4}: fwd {tf:7.1f} us ({flop/tf/1e6:6.1f} TFLOP/s algorithmic)  bwd {tb:7.1f} us")
