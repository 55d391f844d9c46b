"""Roofline table of the non-headline kernels from the artefacts of tools/round2_measure.sh:
event times (run_secondary.json), rocprofv3 kernel-trace averages and two PMC passes.  Writes profiles/<tag>_secondary_roofline.json.

    python tools/secondary_roofline.py gpurun_out/round2 r02
"""
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NZ, W, D = 128, 64, 5
F = D * (2 * NZ * NZ + 2 * (NZ // 2 * W + W * W + W * NZ))          # 327 680 FLOP / sample: one pass over the stack's matrices
STASH = 4 * (2 * 1024 + 2 * 2 * 64) / 32                              # bytes / sample / block of the activation stash (HT = WT = 2)
PEAK_BF16, PEAK_FP32, PEAK_HBM = 2516.6, 157.3, 8000.0
ev = json.loads(open(os.path.join(src, "run_secondary.json")).read().strip().splitlines()[-1])
rows = {B: list(csv.DictReader(open(glob.glob(os.path.join(src, f"prof_secondary_{B}", "*kernel_trace.csv"))[0]))) for B in (100, 65536)}


def match(name, key):
    return all(k in name for k in key.split("&"))


def pmc(passdir, counter, key, B):
    f = max(glob.glob(os.path.join(src, f"{passdir}_{B}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and match(r["Kernel_Name"], key)]
    return sum(v) / len(v) if v else None


def trace(key, B):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[B] if match(r["Kernel_Name"], key))
    return (d[len(d) // 2], len(d)) if d else (None, 0)


# op -> (dominant kernel by size, algorithmic FLOP / sample, algorithmic HBM bytes / sample, pipe)
TABLE = {
    "reverse": ({100: "lsnf_small3_rev_kernel", 65536: "lsnf_rev3_kernel"}, F, 8 * NZ + 8, "bf16"),
    # the stash-writing instantiation of the pipelined forward (template flag true): + block outputs + stash
    "forward + stash (pipelined)": ({100: "lsnf_small3_fwd_kernel", 65536: "lsnf_fwd3q_kernel<2, 8, 2, 1>"}, F, 8 * NZ + 8 + (D - 1) * 4 * NZ + D * STASH, "bf16"),
    # (kernel names carry the template flags: "false>" = no parameter-gradient dump; the Langevin update runs the same kernel)
    "backward_z_from_stash (+ Langevin update)": ({100: "lsnf_small3_bwd_kernel&false>", 65536: "lsnf_bwd3_kernel&false>"}, F, 4 * NZ * 2 + D * (2 * NZ + STASH), "bf16"),
    # (65 536 rows: tiled dump, g_v as its first half only)
    "backward_from_stash + g dump": ({100: "lsnf_small3_bwd_kernel&true>", 65536: "lsnf_bwd3_kernel&true>"}, F, 4 * NZ * 2 + D * (2 * NZ + STASH) + D * 4 * (NZ // 2 + 2 * W + 2 * (NZ // 2)), "bf16"),
    # (one kernel serves forward / + stash / + h dump: the median is over the driver's mix of the three)
    "forward + stash + h dump (65 536 rows: the pipelined kernel, tiled h dump; 100 rows: the latency kernel's mix of the three forms)": ({100: "lsnf_small3_fwd_kernel", 65536: "lsnf_fwd3q_kernel<2, 8, 2, 2>"}, F, 8 * NZ + 8 + (D - 1) * 4 * NZ + D * STASH + D * 8 * W, "bf16"),
    "restash": ({100: "lsnf_small3_restash_kernel", 65536: "lsnf_small3_restash_kernel"}, D * 2 * (NZ // 2 * W + W * W + W * NZ // 2), D * (2 * NZ + STASH), "bf16"),
    # (65 536 rows: lsnf_params3.hip on the bf16 pipe, h2 read once; 100 rows: the fp32-MFMA kernel, h2 read per task)
    "batch_contraction": ({100: "lsnf_tn_gemm_kernel", 65536: "lsnf_contract_x3_kernel"}, F, D * 4 * (2 * NZ + NZ // 2 + 4 * W + 2 * (NZ // 2) + NZ // 2), "bf16 (65536) / fp32 (100)"),
    "unfold": ({100: "lsnf_unfold_kernel", 65536: "lsnf_unfold_kernel"}, 0, 0, "fp64 vector"),
}
out = {"geometry": "C3: nz=128 f_width=64 f_depth=5, default arithmetic bf16x3", "flop_per_sample_one_pass": F,
       "peaks": {"bf16_mfma_tflops": PEAK_BF16, "fp32_mfma_tflops": PEAK_FP32, "hbm_gbs": PEAK_HBM},
       "note": "alg_* = algorithmic work / trace median of the dominant kernel; frac_pipe = algorithmic TFLOP/s / dense peak of the pipe the "
               "kernel's GEMMs run on (an fp32-accurate product costs 6 bf16 MFMAs: frac_pipe <= 1/6 there); frac_hbm = algorithmic GB/s / 8 TB/s; "
               "pmc_* from separate rocprofv3 --pmc passes of the same driver (HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)",
       "event_times_us": ev, "kernels": {}}
for op, (kern, flop, byts, pipe) in TABLE.items():
    for B in (100, 65536):
        key = kern[B]
        med, n = trace(key, B)
        if med is None:
            continue

        e = {"op": op, "B": B, "kernel": key, "trace_median_us": med, "dispatches": n,
             "alg_gflop": flop * B / 1e9, "alg_mbytes": byts * B / 1e6}
        if flop:
            tf = flop * B / med / 1e6
            e["alg_tflops"] = tf
            on_bf16 = pipe == "bf16" or (pipe.startswith("bf16 (65536)") and B == 65536)
            e["frac_pipe"] = tf / (PEAK_BF16 if on_bf16 else PEAK_FP32)
            e["pipe"] = ("bf16" if on_bf16 else "fp32") + " MFMA"
        if byts:
            e["alg_gbs"] = byts * B / med / 1e3
            e["frac_hbm"] = e["alg_gbs"] / PEAK_HBM
        for name, pd, c in (("pmc_insts_valu", "pmc_s1", "SQ_INSTS_VALU"), ("pmc_insts_mfma", "pmc_s1", "SQ_INSTS_MFMA"),
                            ("pmc_fetch_kb", "pmc_s2", "FETCH_SIZE"), ("pmc_write_kb", "pmc_s3", "WRITE_SIZE")):
            try:
                v = pmc(pd, c, key, B)
            except Exception:
                v = None
            if v is not None:
                e[name] = v
        if "pmc_fetch_kb" in e and "pmc_write_kb" in e:
            e["pmc_hbm_mbytes"] = (2 * e["pmc_fetch_kb"] + e["pmc_write_kb"]) * 1024 / 1e6
        out["kernels"][f"{op} B={B}"] = e
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_secondary_roofline.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k:36s} {e['kernel']:30s} {e['trace_median_us']:9.1f} us  " + (f"{e.get('alg_tflops', 0):7.1f} TF ({e.get('frac_pipe', 0):.3f} of {e.get('pipe', '')})  " if e.get("alg_tflops") else "") +
          (f"{e.get('alg_gbs', 0):7.0f} GB/s ({e.get('frac_hbm', 0):.3f})" if e.get("alg_gbs") else ""))
