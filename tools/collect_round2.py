"""Copies what tools/round2_measure.sh left under gpurun_out/round2/ into profiles/ under a tag (default r02) and re-derives
the summaries: kernel-only averages from the trace of the bench command, traffic.json (bf16x3 = the headline mode), the
secondary roofline table.   python tools/collect_round2.py [tag]"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else "round2"          # gpurun_out/<src>: round2 (tools/round2_measure.sh), round3 (tools/round3_measure.sh)
R, P = os.path.join(ROOT, "gpurun_out", src), os.path.join(ROOT, "profiles")
last = lambda f: open(f).read().strip().splitlines()[-1] + "\n"
for name in ("bench", "bench_fp32", "bench_fp16x2", "bench_driver_args"):
    open(os.path.join(P, f"{tag}_{name}.json"), "w").write(last(os.path.join(R, f"{name}.json")))
shutil.copy(os.path.join(R, "prof", "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
os.makedirs(os.path.join(P, f"{tag}_pmc"), exist_ok=True)
for p in ("p1", "p2", "p3", "p4"):
    shutil.copy(max(glob.glob(os.path.join(R, "pmc", p, "*", "*_counter_collection.csv")), key=os.path.getmtime), os.path.join(P, f"{tag}_pmc", f"{p}_counter_collection.csv"))
os.makedirs(os.path.join(P, f"{tag}_pmc_secondary"), exist_ok=True)
for B in (100, 65536):
    for p in ("pmc_s1", "pmc_s2", "pmc_s3"):
        shutil.copy(max(glob.glob(os.path.join(R, f"{p}_{B}", "*", "*_counter_collection.csv")), key=os.path.getmtime),
                    os.path.join(P, f"{tag}_pmc_secondary", f"{p}_B{B}_counter_collection.csv"))
    shutil.copy(glob.glob(os.path.join(R, f"prof_secondary_{B}", "*kernel_stats.csv"))[0], os.path.join(P, f"{tag}_secondary_B{B}_kernel_stats.csv"))
shutil.copy(os.path.join(R, "secondary.json"), os.path.join(P, f"{tag}_secondary.json"))
open(os.path.join(P, f"{tag}_run_secondary.json"), "w").write(last(os.path.join(R, "run_secondary.json")))
rows = list(csv.DictReader(open(os.path.join(R, "prof", "bench_kernel_trace.csv"))))
out = {}
for key, name in (("lsnf_fwd3q", "lsnf_fwd3q_kernel<2, 8, 2>"), ("lsnf_fwd3b", "lsnf_fwd3b_kernel<Fwd3Cfg<2,2>, 8>"), ("lsnf_fwd2h", "lsnf_fwd2h_kernel<Fwd3Cfg<2,2>, 8>"),
                  ("lsnf_fwd_kernel", "lsnf_fwd_kernel<FwdCfg<2,2>, 8>")):
    rr = sorted((r for r in rows if key in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    d_all = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rr]
    d = [x for x in d_all if x > 20.0]                 # full runs (the fp16 mode also queues early-exit fix-up launches of fwd3b)
    if len(d) < 200:
        continue
    # bench.py's kernel-only loops run last, single stream: fp32, fp16x2 (+ its fix-up), then the measured mode
    out[name] = {"dispatches_in_trace": len(d), "kernel_only_loop_last_200_avg_us": sum(d[-200:]) / 200, "min_us": min(d[-200:]), "max_us": max(d[-200:])}
out["note"] = ("per-dispatch durations from rocprofv3 --kernel-trace of `python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline` (default --math "
               "bf16x3); the timed steps rotate over 3 HIP streams, so their kernels overlap and the all-dispatch average of the kernel_stats csv is NOT "
               "a kernel duration; bench.py's roofline uses its single-stream kernel-only loop (the last 200 dispatches of each kernel), which is what "
               "is averaged here")
json.dump(out, open(os.path.join(P, f"{tag}_kernel_only_from_trace.json"), "w"), indent=1)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), os.path.join(R, "pmc"), "bf16x3"], cwd=ROOT, check=True)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "secondary_roofline.py"), R, tag], cwd=ROOT, check=True)
for n in (8, 4, 2):                                            # round 3: one GPU's shard of an N-GPU step
    f = os.path.join(R, f"shard_of_{n}.json")
    if os.path.exists(f):
        open(os.path.join(P, f"{tag}_shard_of_{n}.json"), "w").write(last(f))
if os.path.exists(os.path.join(R, "shard_times.txt")):
    shutil.copy(os.path.join(R, "shard_times.txt"), os.path.join(P, f"{tag}_shard_times.txt"))
if os.path.exists(os.path.join(R, "prof_shard8", "shard8_kernel_stats.csv")):
    shutil.copy(os.path.join(R, "prof_shard8", "shard8_kernel_stats.csv"), os.path.join(P, f"{tag}_shard_of_8_kernel_stats.csv"))
b = json.loads(last(os.path.join(R, "bench.json")))
print(tag, "value %.4g" % b["value"], "ms/step %.4f" % b["ms_per_step"], "kernel_ms %.4f" % b["roofline"]["kernel_ms"], "frac %.4f" % b["roofline"]["frac"],
      "| trace", {k: round(v["kernel_only_loop_last_200_avg_us"], 1) for k, v in out.items() if k != "note"})
