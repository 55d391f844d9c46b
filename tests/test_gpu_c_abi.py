"""GPU: a plain-C program (tests/c_abi_smoke.c, hipcc, no Python/torch) drives the library through include/lsnf_flow.h
and checks a closed-form flow -- the C ABI is a real language-neutral boundary, not a torch extension."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_caller(tmp_path, gpu_device):
    exe = tmp_path / "c_abi_smoke"
    pkg = os.path.join(ROOT, "latent-space-normalizing-flow_amd")
    cmd = ["/opt/rocm/bin/hipcc", "-x", "c", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", str(exe),
           "-L", pkg, "-llsnf_flow", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("-> ok") == 2
