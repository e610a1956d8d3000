#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory (kernel stats + separate PMC passes) into profiles/.

usage: tools/summarize_prof.py <gpurun_out/prof_dir> <profiles/rNN> <tag> <traffic_key>
"""
import collections
import csv
import json
import os
import shutil
import sys

base, outdir, tag, key = sys.argv[1:5]
os.makedirs(outdir, exist_ok=True)
shutil.copy(os.path.join(base, "trace", "bench_kernel_stats.csv"), os.path.join(outdir, f"rocprofv3_kernel_stats_{tag}.csv"))
line = [l for l in open(os.path.join(base, "bench_trace.log")) if l.startswith("{")][-1]
open(os.path.join(outdir, f"bench_line_under_rocprof_{tag}.json"), "w").write(line)
bench = json.loads(line)
KN = bench["roofline"]["kernel"]                 # symbol of the headline variant
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py   (PMC: separate runs with "
                   "--pmc <counters> only, bench.py --steps 3 --warmup 1 --no-cpu-baseline)",
        "workload": bench["config"]["workload"], "kernels": {}}
stats = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(os.path.join(base, "trace", "bench_kernel_stats.csv")))}
summ["kernel_trace"] = {k: {"calls": int(v["Calls"]), "avg_ms": float(v["AverageNs"]) / 1e6, "min_ms": float(v["MinNs"]) / 1e6,
                            "max_ms": float(v["MaxNs"]) / 1e6} for k, v in stats.items() if "fz" in k}
# steady-state launches only (the trace CSV has every dispatch)
tr = [r for r in csv.DictReader(open(os.path.join(base, "trace", "bench_kernel_trace.csv"))) if r["Kernel_Name"] == KN]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
steps = bench["steps"]
summ["kernel_trace"][KN]["timed_region_avg_ms"] = sum(durs[-steps:]) / steps
summ["kernel_trace"][KN]["bench_event_avg_ms"] = bench["roofline"]["avg_launch_ms"]
for d in sorted(os.listdir(base)):
    f = os.path.join(base, d, "bench_counter_collection.csv")
    if not d.startswith("pmc_") or not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "fz" not in n:
            continue
        k = summ["kernels"].setdefault(n, {"VGPR_Count": r["VGPR_Count"], "SGPR_Count": r["SGPR_Count"], "Grid_Size": r["Grid_Size"],
                                           "Workgroup_Size": r["Workgroup_Size"], "LDS_Block_Size": r["LDS_Block_Size"], "counters": {}})
        c = k["counters"].setdefault(r["Counter_Name"], {"per_launch": [], "ms": []})
        c["per_launch"].append(float(r["Counter_Value"]))
        c["ms"].append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
for k in summ["kernels"].values():
    for c in k["counters"].values():
        c["mean"] = sum(c["per_launch"]) / len(c["per_launch"])
PK = KN if KN in summ["kernels"] else sorted(k for k in summ["kernels"] if k.startswith("fz_block_kernel_p"))[0]
summ["pmc_kernel"] = PK   # variant the PMC passes ran (plan selection off: the library default)
blk = summ["kernels"][PK]["counters"]
cp = summ["kernels"]["fz::fz_copy_kernel"]["counters"]
copy_bytes = bench["config"]["streams_per_gpu"] * bench["config"]["block_samples"] * 4
b_alg = bench["roofline"]["algorithmic_bytes_per_launch"]
rf = copy_bytes / (cp["FETCH_SIZE"]["mean"] * 1024)
wf = copy_bytes / (cp["WRITE_SIZE"]["mean"] * 1024)
traffic = blk["FETCH_SIZE"]["mean"] * 1024 * 2 + blk["WRITE_SIZE"]["mean"] * 1024
summ["hbm_traffic"] = {
    "method": "MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE from separate --pmc passes, unit KiB; gfx950 "
              "FETCH_SIZE counts 1/2 of a coalesced streaming read -> x2; calibrated in the same runs on fz_copy_kernel "
              "(exactly %d bytes each way)" % copy_bytes,
    "calibration_copy_kernel": {"read_factor": round(rf, 4), "write_factor": round(wf, 4)},
    PK: {"FETCH_SIZE_KiB": blk["FETCH_SIZE"]["mean"], "WRITE_SIZE_KiB": blk["WRITE_SIZE"]["mean"],
                        "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": b_alg,
                        "traffic_over_algorithmic": round(traffic / b_alg, 5)}}
if "SQ_INSTS_VALU" in blk:
    waves = blk["SQ_WAVES"]["mean"]
    summ["valu"] = {"SQ_WAVES": waves, "SQ_INSTS_VALU_per_launch": blk["SQ_INSTS_VALU"]["mean"],
                    "valu_insts_per_wave_per_sample": blk["SQ_INSTS_VALU"]["mean"] / waves / bench["config"]["block_samples"],
                    "SQ_BUSY_CYCLES": blk["SQ_BUSY_CYCLES"]["mean"], "SQ_WAVE_CYCLES": blk["SQ_WAVE_CYCLES"]["mean"],
                    "SQ_WAIT_ANY": blk["SQ_WAIT_ANY"]["mean"], "SQ_WAIT_INST_ANY": blk["SQ_WAIT_INST_ANY"]["mean"]}
json.dump(summ, open(os.path.join(outdir, f"rocprofv3_pmc_summary_{tag}.json"), "w"), indent=1)
tp = os.path.join(os.path.dirname(outdir.rstrip("/")), "pmc_traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[key] = traffic
t["_source"] = "profiles/r01/rocprofv3_pmc_summary_*.json (tools/summarize_prof.py)"
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps(summ["kernel_trace"], indent=1))
print(json.dumps(summ["hbm_traffic"], indent=1))
print(json.dumps(summ.get("valu"), indent=1))
