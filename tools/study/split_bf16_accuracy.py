"""Study (CPU, no GPU): accuracy of error-free bf16 splitting on the flow's GEMMs.

Every GEMM of the forward path is evaluated as a sum of bf16 x bf16 products with fp32 accumulation
(what v_mfma_f32_32x32x16_bf16 does), operands split as x = x1 + x2 + x3 (each bf16, round-to-nearest):
  6 terms: (1,1) (1,2) (2,1) (2,2) (1,3) (3,1)      3 terms: (1,1) (1,2) (2,1) with a 2-way split
and the resulting log-prob is compared with a float64 evaluation, next to plain fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flow_oracle as O

torch.set_num_threads(8)

def split(x, n):
    parts, r = [], x.clone()
    for _ in range(n):
        p = r.bfloat16().float()
        parts.append(p); r = r - p
    return parts

def split_h(x, n, rtz=False):
    """error-free fp16 split (RNE, or round-toward-zero like v_cvt_pkrtz_f16_f32); fp16 subnormals kept"""
    parts, r = [], x.clone()
    for _ in range(n):
        if rtz:
            h = r.half().float()
            over = h.abs() > r.abs()
            h = torch.where(over, torch.nextafter(h.half(), torch.zeros_like(h).half()).float(), h)
        else:
            h = r.half().float()
        parts.append(h); r = r - h
    return parts

def mm(x, w, mode):
    if mode == "fp32": return x @ w
    if mode == "fp64": return (x.double() @ w.double())
    if mode.startswith("fp16x2"):
        # operands scaled by powers of two (exact) so that the low parts stay in fp16's normal range where possible
        sw = 2.0 ** torch.floor(torch.log2(1.0 / w.abs().max())).item() * 16.0 if "scaled" in mode else 1.0
        sx = 2.0 ** torch.floor(torch.log2(1.0 / x.abs().max().clamp_min(1e-30))).item() * 16.0 if "scaled" in mode else 1.0
        xs, ws = split_h(x * sx, 2, "rtz" in mode), split_h(w * sw, 2, "rtz" in mode)
        terms = [(0,0),(0,1),(1,0),(1,1)] if "4t" in mode else [(0,0),(0,1),(1,0)]
        acc = torch.zeros(x.shape[0], w.shape[1])
        for i, j in reversed(terms):
            acc = acc + xs[i] @ ws[j]
        return acc / (sx * sw)
    n, terms = (3, [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]) if mode == "bf16x6" else \
               (3, [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0),(1,2),(2,1),(2,2)]) if mode == "bf16x9" else (2, [(0,0),(0,1),(1,0)])
    xs, ws = split(x, n), split(w, n)
    acc = torch.zeros(x.shape[0], w.shape[1])
    for i, j in reversed(terms):           # small terms first
        acc = acc + xs[i] @ ws[j]
    return acc

def flow_ll(p, z, mode):
    dt = torch.float64 if mode == "fp64" else torch.float32
    depth = O.depth_of(p)
    x = z.to(dt); ell = torch.zeros(z.shape[0], dtype=dt)
    half = z.shape[1] // 2
    for i in range(depth):
        pre = O.block_prefix(i)
        g = lambda k: p[pre + k].to(torch.float64)
        ea = torch.exp(3 * g("actnorm.logs")).reshape(-1)
        W = g("invertible_1x1_conv.w")
        Wa = (ea[:, None] * W); ca = ((g("actnorm.b").reshape(-1) * ea) @ W)
        e1 = torch.exp(3 * g("f.fc_1.actnorm.logs")).reshape(-1); W1 = g("f.fc_1.w") * e1; c1 = g("f.fc_1.actnorm.b").reshape(-1) * e1
        e2 = torch.exp(3 * g("f.fc_2.actnorm.logs")).reshape(-1); W2 = g("f.fc_2.w") * e2; c2 = g("f.fc_2.actnorm.b").reshape(-1) * e2
        e3 = torch.exp(3 * g("f.fc_zeros.logs")).reshape(-1); W3 = g("f.fc_zeros.w") * e3; c3 = g("f.fc_zeros.b").reshape(-1) * e3
        ell = ell + (3 * g("actnorm.logs").sum() + torch.slogdet(W)[1]).to(dt)
        v = mm(x, Wa.to(dt), mode).to(dt) + ca.to(dt)
        v1, v2 = v[:, :half], v[:, half:]
        h1 = torch.relu(mm(v1, W1.to(dt), mode).to(dt) + c1.to(dt))
        h2 = torch.relu(mm(h1, W2.to(dt), mode).to(dt) + c2.to(dt))
        hh = mm(h2, W3.to(dt), mode).to(dt) + c3.to(dt)
        t, pp = hh[:, 0::2], hh[:, 1::2] + 2.0
        sg = torch.sigmoid(pp)
        x = torch.cat([v1, (v2 + t) * sg], 1)
        ell = ell + torch.log(sg).sum(1)
    return (-0.5 * (x ** 2).sum(1) + float(np.log(2 * np.pi)) + ell)

for (nz, w, scale, tag) in [(128, 64, 0.05, "C3 init-like"), (128, 64, 0.12, "C3 trained-like"), (100, 128, 0.12, "C5 trained-like")]:
    p = O.init_params(nz, w, 5, seed=3, fcz_std=scale, all_std=0.04 if scale > 0.1 else 0.0)
    z = torch.randn(4096, nz, generator=torch.Generator().manual_seed(1)) * (2.0 if scale > 0.1 else 1.0)
    ref = flow_ll(p, z, "fp64")
    print(tag, "| |ll| median", float(ref.abs().median()))
    for mode in ("fp32", "bf16x9", "bf16x6", "bf16x3", "fp16x2", "fp16x2_scaled", "fp16x2_scaled_4t", "fp16x2_scaled_rtz"):
        ll = flow_ll(p, z, mode).double()
        rel = ((ll - ref).abs() / ref.abs().clamp_min(1.0))
        print(f"   {mode:18s} max rel err {rel.max().item():.3e}   median {rel.median().item():.3e}")
