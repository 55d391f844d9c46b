"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("langevin", "netg")))


def load_golden(name):
    """Returns (params dict keyed like the reference state_dict, incl. the '.bias' aliases;
    dict of the remaining arrays)."""
    raw = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    params, rest = {}, {}
    for k in raw.files:
        if k.startswith("sd/"):
            params[k[3:]] = torch.from_numpy(raw[k].copy())
        else:
            rest[k] = raw[k]
    for k in list(params):
        if k.endswith("actnorm.b"):
            params[k + "ias"] = params[k]
    return params, rest


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(params=["latency-kernels", "latency-kernels-bf16x3", "throughput-kernels", "throughput-kernels-bf16x3",
                        "throughput-kernels-bf16x3_phased", "throughput-kernels-fp16x2"])
def kernels(request):
    """Every parity test runs on both kernel families (normally selected by batch size, lsnf_set_small_batch_max) and,
    for the throughput family, with both arithmetic modes of the forward's GEMMs (lsnf_set_math_mode: fp32 MFMA, or
    the error-free three-way bf16 split on the bf16 matrix pipe -- software-pipelined (default) or phase-separated -- or
    the two-way fp16 split)."""
    import lsnf_amd
    prev = lsnf_amd.flow.set_small_batch_max(1 << 30 if request.param.startswith("latency-kernels") else 0)
    prev_math = lsnf_amd.flow.set_math_mode(lsnf_amd.flow.MATH_BF16X3 if request.param.endswith("bf16x3")
                                            else lsnf_amd.flow.MATH_BF16X3_PHASED if request.param.endswith("bf16x3_phased")
                                            else lsnf_amd.flow.MATH_FP16X2 if request.param.endswith("fp16x2")
                                            else lsnf_amd.flow.MATH_FP32)
    yield request.param
    lsnf_amd.flow.set_small_batch_max(prev)
    lsnf_amd.flow.set_math_mode(prev_math)
