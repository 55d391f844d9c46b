"""GPU parity, randomised: tools/fuzz_parity.py's sweep (random nz / width / depth, batch sizes on the kernel-family and
tile boundaries, every entry point -- forward with and without the activation stash, backward w.r.t. z, the Langevin
step with tensor and in-kernel Philox noise, reverse, round trip, parameter gradients -- in the three arithmetic modes)
against the float64 oracle.  A fixed seed keeps the run reproducible; `python tools/fuzz_parity.py N SEED` runs more."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_geometries_match_oracle(gpu_device):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    small = [b for b in fuzz_parity.BATCHES if b <= 4097]
    n, bad = fuzz_parity.run(16, seed=11, batches=small)
    assert n > 500 and not bad, bad[:5]
    n, bad = fuzz_parity.run(3, seed=12, batches=[16384, 16385, 32769])      # both sides of the family / wave-count switches
    assert n > 60 and not bad, bad[:5]
