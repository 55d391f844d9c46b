"""GPU, two ranks on ONE card (SURVEY 8e while no multi-GPU node is available): two fresh child processes, gloo between
them, each running the HIP path on its slab of the batch -- sharded log-prob with in-kernel sums and the bucket-1 reducer,
a sharded Langevin step with the in-kernel noise keyed by global rows, slab MLE gradients reduced as one bucket -- against
the same calls on the whole batch by one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (batch, nz, width): slab and batch inside the latency family (16-row and 32-row workgroups); slab in the latency family,
# batch in the throughput family (the forward kernels differ: rows agree bit for bit in z, to rounding in the log-prob)
@pytest.mark.parametrize("B,nz,width,same_family", [(6000, 128, 64, True), (30001, 128, 64, False), (777, 100, 64, True)])
def test_two_ranks_one_gpu_match_single_rank(gpu_device, tmp_path, B, nz, width, same_family):
    out = tmp_path / "multirank.json"
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_gpu_worker.py"), str(out), str(B), str(nz), str(width)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r = json.loads(out.read_text())
    assert r["world"] == 2 and r["rows_total"] == B
    # rows: the slab results ARE the single-rank results
    assert r["z1_bitwise"] and r["z_new_max_abs"] <= 2e-6
    if same_family:
        assert r["ll_bitwise"] and r["z_new_bitwise"] and r["gf_norm_bitwise"]
    else:
        assert r["ll_max_rel"] <= 2e-6
    # sums: in-kernel partial sums + one all-reduce vs one launch over the whole batch
    for a, b in ((r["sum_ll"], r["sum_ll_single"]), (r["sum_ld"], r["sum_ld_single"])):
        assert abs(a - b) <= (1e-9 if same_family else 1e-7) * abs(b)
    assert abs(r["sum_ll"] - r["sum_ll_fp64_of_rows"]) <= (1e-9 if same_family else 1e-7) * abs(r["sum_ll_fp64_of_rows"])
    # gradients: fp32 atomics in the batch contraction -- the run-to-run spread bound of the single-rank path (2e-6), with room
    # for the different summation order of two slabs.  Across kernel families the ReLU masks come from two forwards whose
    # pre-activations differ in their last bits: among tens of thousands of rows a handful sit within rounding of a kink and
    # pick the neighbouring linear piece (batch-summed gradients allow 5e-4, measured 1.2e-4)
    assert r["n_grad_tensors"] == 60 and r["grad_max_rel"] <= (2e-5 if same_family else 5e-4)
