"""Turn the PMC passes of tools/pmc_fwd.sh into profiles/traffic.json (read by bench.py).
HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters; gfx950 correction of
MI355X_MICROARCH.md 'HBM': FETCH_SIZE reports half the bytes of a wide coalesced read)."""
import csv, glob, json, sys, os
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
math = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"      # which arithmetic mode the PMC run profiled (LSNF_MATH)
KEY = {"bf16x3": "lsnf_fwd3q_kernel", "fp16x2": "lsnf_fwd2h_kernel"}.get(math, "lsnf_fwd_kernel")   # the mode's dominant kernel
def mean(counter, p):
    f = max(glob.glob(os.path.join(root, p, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest (older runs may linger)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and KEY in r["Kernel_Name"]]
    return sum(v) / len(v)
fetch_kb, write_kb = mean("FETCH_SIZE", "p3"), mean("WRITE_SIZE", "p4")
out = {"kernel": {"bf16x3": "lsnf_fwd3q_kernel<2, 8, 2>", "fp16x2": "lsnf_fwd2h_kernel<Fwd3Cfg<2,2>, 8>"}.get(math, "lsnf_fwd_kernel<FwdCfg<2,2>, 8 waves>"), "workload": "nz=128 w=64 depth=5 B=65536",
       "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
       "algorithmic_bytes_per_launch": 1032 * 65536,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read side doubled (gfx950 correction)"}
allm = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
allm[math] = out
json.dump(allm, open("profiles/traffic.json", "w"), indent=1)
print(out)
