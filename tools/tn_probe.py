"""GPU experiment: where the batch contraction of lsnf_params.hip spends its time at B = 65 536 (env knobs LSNF_TN_ABL /
LSNF_TN_CHUNK / LSNF_TN_PLAIN, one subprocess each)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, %r)
import lsnf_amd
from lsnf_amd import flow
dev = torch.device("cuda:0")
hps = types.SimpleNamespace(f_n_levels=1, f_depth=5, f_flow_permutation=2, f_width=64, f_flow_coupling=1)
torch.manual_seed(1); np.random.seed(1)
net = lsnf_amd._netF(hps, nz=128).to(dev); plan = net._plan(); params = [p.detach() for p in net._param_list()]
B = 65536
z = torch.randn(B, 128, device=dev)
act = flow.new_act_saved(plan, B, dev); ws = flow.new_params_workspace(plan, B, dev)
z1, _, _, saved = flow.forward(plan, z, want_ll=False, save_for_backward=True, act_saved=act, params_ws=ws)
def run(): flow.backward_params(plan, params, z, z1, saved, ll_scale=-1.0 / B, reuse_buffers=True, act_saved=act, workspace=ws)
for _ in range(10): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): run()
e1.record(); torch.cuda.synchronize()
print("%%.1f us" %% (e0.elapsed_time(e1) / 30 * 1e3))
''' % ROOT
for name, env in (("lds kernel", {}), ("lds, no atomics (chunk 0 only)", {"LSNF_TN_ABL": "1"}), ("lds, no MFMA", {"LSNF_TN_ABL": "2"}), ("lds, no global loads", {"LSNF_TN_ABL": "4"}),
                  ("lds, chunk 512", {"LSNF_TN_CHUNK": "512"}), ("lds, chunk 2048", {"LSNF_TN_CHUNK": "2048"}), ("lds, chunk 4096", {"LSNF_TN_CHUNK": "4096"}),
                  ("plain kernel", {"LSNF_TN_PLAIN": "1"}), ("plain, chunk 512", {"LSNF_TN_PLAIN": "1", "LSNF_TN_CHUNK": "512"})):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True, timeout=200)
    print(f"{name:36s} backward_params (fast path, total) {r.stdout.strip() or r.stderr[-300:]}", flush=True)
