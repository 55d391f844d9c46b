"""Generator of the latent-space model, `_netG` -- the OTHER network inside every Langevin step.

Not a hand-written kernel: the generator is a stack of transposed convolutions, i.e. vendor-library work (MIOpen via
PyTorch-ROCm).  It is mirrored here so that the reference's whole `from model import _netG, _netF` (train.py:32) can
be switched and so that the training-step measurements of examples/train_synthetic.py run the reference's own
generator shapes.  Same constructor argument (an attribute bag with dataset, nz, ngf, nc, g_activation,
g_activation_leak, g_batchnorm), same `state_dict` keys (`gen.<i>.weight` ...) and layer order as reference
model.py:48-157 -- built from one table per dataset instead of four written-out stacks.

`tune()` applies the two MIOpen-side settings that matter for the Langevin loop (a forward + an input-gradient of the
same shapes, K times per iteration): channels-last weights/activations and MIOpen's find mode (`cudnn.benchmark` on
ROCm), which searches the solver once per shape."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

# dataset -> [(output channels as a multiple of ngf, or "nc", kernel, stride, padding), ...]   (model.py:52-150)
_UP = (4, 2, 1)          # the doubling layer: kernel 4, stride 2, padding 1
STACKS = {
    "svhn":         [(8, 4, 1, 0), (4, *_UP), (2, *_UP), ("nc", *_UP)],                                   # 1 -> 4 -> 32
    "cifar10":      [(8, 8, 1, 0), (4, *_UP), (2, *_UP), ("nc", 3, 1, 1)],                                # 1 -> 8 -> 32
    "celeba_crop":  [(8, 4, 1, 0), (4, *_UP), (2, *_UP), (1, *_UP), ("nc", *_UP)],                        # 1 -> 4 -> 64
    "celeba_hq256": [(16, 4, 1, 0), (8, *_UP), (4, *_UP), (2, *_UP), (1, *_UP), (1, *_UP), ("nc", *_UP)],  # 1 -> 4 -> 256
}


class _XTanhSoftplus(nn.Module):      # "mish" (model.py:20-25)
    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


class _XSigmoid(nn.Module):           # "swish" (model.py:27-32)
    def forward(self, x):
        return x * torch.sigmoid(x)


def _activation(name: str, leak: float) -> nn.Module:
    table = {"lrelu": lambda: nn.LeakyReLU(leak), "gelu": nn.GELU, "mish": _XTanhSoftplus, "swish": _XSigmoid}
    if name not in table:
        raise KeyError(name)
    return table[name]()


class _netG(nn.Module):
    """z (B, nz, 1, 1) -> image (B, nc, H, W) in [-1, 1].  `args`: see the module docstring."""

    def __init__(self, args):
        super().__init__()
        if args.dataset not in STACKS:
            raise ValueError(args.dataset)                      # as model.py:152
        act = _activation(args.g_activation, getattr(args, "g_activation_leak", 0.2))   # ONE shared instance, as the reference
        bn = bool(args.g_batchnorm)
        layers, cin = [], args.nz
        spec = STACKS[args.dataset]
        for i, (mult, k, s, p) in enumerate(spec):
            last = i == len(spec) - 1
            cout = args.nc if mult == "nc" else args.ngf * mult
            layers.append(nn.ConvTranspose2d(cin, cout, k, s, p, bias=True if last else not bn))
            if last:
                layers.append(nn.Tanh())
            else:
                layers += [nn.BatchNorm2d(cout) if bn else nn.Identity(), act]
            cin = cout
        self.gen = nn.Sequential(*layers)

    def forward(self, z):
        return self.gen(z)

    def tune(self, channels_last: bool = True, find_mode: bool = True) -> "_netG":
        """MIOpen-side settings for the Langevin loop (see the module docstring); returns self."""
        if find_mode:
            torch.backends.cudnn.benchmark = True
        if channels_last:
            self.to(memory_format=torch.channels_last)
        return self


def langevin_grad_g(netG: nn.Module, z: torch.Tensor, x: torch.Tensor, sigma: float):
    """train.py:312-314: x_hat = G(z); g_log_lkhd = |x_hat - x|^2 / (2 sigma^2); returns (d g_log_lkhd / dz, g_log_lkhd)."""
    z = z.detach().requires_grad_(True)
    g_log_lkhd = F.mse_loss(netG(z), x, reduction="sum") / (2.0 * sigma * sigma)
    return torch.autograd.grad(g_log_lkhd, z)[0], g_log_lkhd.detach()
