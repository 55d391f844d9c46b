"""Child process of tests/test_gpu_module.py::test_generator_mirror_on_device_matches_reference_golden: `lsnf_amd._netG` on the GPU
(stock PyTorch-ROCm / MIOpen; plain and tuned = channels-last + find mode) and `netg.langevin_grad_g` against the reference's
`_netG` golden vectors (tests/golden/netg_variants.npz), every dataset variant.  Prints `generator_mirror_ok <cases>` and exits 0."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsnf_amd                          # noqa: E402,F401
from lsnf_amd import netg                # noqa: E402

dev = torch.device("cuda:0")
raw = np.load(os.path.join(ROOT, "tests", "golden", "netg_variants.npz"), allow_pickle=False)
tags = sorted({k.split("/")[0] for k in raw.files})
assert len(tags) == 5
cases = 0
for tuned in (False, True):
    for tag in tags:
        ds, act, bn = tag.rsplit("_", 2)
        size, nz, ngf, B, sub = (int(v) for v in raw[f"{tag}/meta"])
        args = types.SimpleNamespace(dataset=ds, nz=nz, ngf=ngf, nc=3, g_activation=act, g_activation_leak=0.2,
                                     g_batchnorm=bn == "bn1")
        net = netg._netG(args).eval()
        net.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(raw[k]) for k in raw.files if k.startswith(tag + "/sd/")},
                            strict=True)
        net = net.to(dev)
        if tuned:
            net.tune()
        z = torch.from_numpy(raw[f"{tag}/z"]).to(dev)
        b, c, i, j = np.meshgrid(np.arange(B), np.arange(3), np.arange(size), np.arange(size), indexing="ij")
        x = torch.from_numpy(np.tanh(np.sin(0.37 * i + 0.91 * j + 1.7 * c + 2.3 * b)).astype(np.float32)).to(dev)
        with torch.no_grad():
            x_hat = net(z)
        assert (x_hat[:, :, ::sub, ::sub].cpu() - torch.from_numpy(raw[f"{tag}/x_hat"])).abs().max().item() <= 2e-5, tag
        zg, gl = netg.langevin_grad_g(net, z, x, 0.3)
        ref = torch.from_numpy(raw[f"{tag}/z_grad_g"])
        assert abs(gl.item() - float(raw[f"{tag}/g_log_lkhd"])) <= 2e-5 * abs(float(raw[f"{tag}/g_log_lkhd"])), tag
        assert (zg.cpu() - ref).norm().item() <= 2e-4 * ref.norm().item(), tag
        cases += 1
torch.backends.cudnn.benchmark = False
print("generator_mirror_ok", cases, flush=True)
