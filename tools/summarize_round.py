#!/usr/bin/env python3
"""Summarise tools/profile_round.sh output into profiles/<round>/ and profiles/pmc_traffic.json.

usage: tools/summarize_round.py <gpurun_out/dir> <profiles/rNN>

Workload keys (bench.py traffic_of): "<kernel symbol>|<workload key>" -> HBM bytes per launch
  = FETCH_SIZE[KiB] * 1024 * read_factor + WRITE_SIZE[KiB] * 1024 * write_factor, the factors calibrated in the same
  batch on fz_copy_kernel, which moves exactly 16 GiB each way (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half of a
  wide coalesced streaming read)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

base, outdir = sys.argv[1:3]
os.makedirs(outdir, exist_ok=True)
WORKLOAD = {"head": "cascade6_1048576x4096_timemajor", "tl": "cascade6_1048576x4096_tile8192", "c2": "cascade6_65536x4096_tile8192", "c2h": "cascade6_32768x4096_tile8192", "c2q": "cascade6_16384x4096_tile8192", "c3": "par4_1048576x4096_tile4096",
            "c3f": "par4f_1048576x4096_timemajor", "c4": "osc6_1048576x4096_tile8192",
            "tm": "cascade6_1048576x4096_timemajor", "sm": "cascade6_1048576x4096_streammajor"}
B_ALG = {"head": 1048576 * (4 * 4096 * 2 + 8 * 14), "tl": 1048576 * (4 * 4096 * 2 + 8 * 14), "c2": 65536 * (4 * 4096 * 2 + 8 * 14), "c2h": 32768 * (4 * 4096 * 2 + 8 * 14), "c2q": 16384 * (4 * 4096 * 2 + 8 * 14), "c3": 1048576 * (4 * 4096 * 5 + 8 * 16),
         "c3f": 1048576 * (4 * 4096 * 2 + 8 * 18), "c4": 1048576 * (4 * 4096 * 2 + 8 * 16 + 4 * 31),
         "tm": 1048576 * (4 * 4096 * 2 + 8 * 14), "sm": 1048576 * (4 * 4096 * 2 + 8 * 14)}


def counters(d):
    """{kernel: {counter: [values per launch]}} of one pass"""
    f = os.path.join(d, "b_counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[r["Kernel_Name"].split("(")[0]]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out


# 1. kernel trace + stats of the default command
shutil.copy(os.path.join(base, "trace", "bench_kernel_stats.csv"), os.path.join(outdir, "rocprofv3_kernel_stats_default_bench.csv"))
line = [l for l in open(os.path.join(base, "bench_trace.log")) if l.startswith("{")][-1]
open(os.path.join(outdir, "bench_line_under_rocprof.json"), "w").write(line)
plain = [l for l in open(os.path.join(base, "bench_plain.log")) if l.startswith("{")][-1]
open(os.path.join(outdir, "bench_line_plain.json"), "w").write(plain)
bench = json.loads(line)
stats = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(os.path.join(base, "trace", "bench_kernel_stats.csv")))}
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ; PMC: separate runs, --pmc <counter> only "
                   "(tools/profile_round.sh)",
        "kernel_trace": {k: {"calls": int(v["Calls"]), "avg_ms": float(v["AverageNs"]) / 1e6, "min_ms": float(v["MinNs"]) / 1e6,
                             "max_ms": float(v["MaxNs"]) / 1e6} for k, v in stats.items() if "fz" in k}}
KN = bench["roofline"]["kernel"]
tr = [r for r in csv.DictReader(open(os.path.join(base, "trace", "bench_kernel_trace.csv"))) if r["Kernel_Name"] == KN]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
# the K timed steps come right after tune + warm-up and before the sustained run: compare the trace's average with the events'
summ["headline_kernel"] = {"symbol": KN, "rocprof_avg_ms_all_calls": sum(durs) / max(len(durs), 1), "calls": len(durs),
                           "bench_event_avg_launch_ms": bench["roofline"]["avg_launch_ms"],
                           "bench_sustained_avg_launch_ms": bench["roofline"].get("sustained", {}).get("avg_launch_ms")}

# 2. PMC traffic
cal = None
for d in sorted(glob.glob(os.path.join(base, "pmc_head_default_FETCH_SIZE"))):
    cf, cw = counters(d), counters(d.replace("FETCH_SIZE", "WRITE_SIZE"))
    k = "fz::fz_copy_kernel"
    if k in cf and k in cw:
        copy_bytes = 1048576 * 4096 * 4
        cal = {"read_factor": copy_bytes / (sum(cf[k]["FETCH_SIZE"]) / len(cf[k]["FETCH_SIZE"]) * 1024),
               "write_factor": copy_bytes / (sum(cw[k]["WRITE_SIZE"]) / len(cw[k]["WRITE_SIZE"]) * 1024), "copy_bytes_each_way": copy_bytes}
if cal is None:
    cal = {"read_factor": 2.0, "write_factor": 1.0, "note": "copy kernel not found in this batch: the factors measured in round 1"}
summ["calibration_on_copy_kernel"] = cal
traffic, rows = {}, []
for d in sorted(glob.glob(os.path.join(base, "pmc_*_FETCH_SIZE"))):
    tag = os.path.basename(d)[4:-len("_FETCH_SIZE")]
    wl = tag.split("_")[0]
    if wl not in WORKLOAD:
        continue
    cf, cw = counters(d), counters(d.replace("FETCH_SIZE", "WRITE_SIZE"))
    for k in cf:
        if not k.startswith("fz_block_kernel") or k not in cw:
            continue
        f = sum(cf[k]["FETCH_SIZE"]) / len(cf[k]["FETCH_SIZE"]) * 1024
        w = sum(cw[k]["WRITE_SIZE"]) / len(cw[k]["WRITE_SIZE"]) * 1024
        t = f * cal["read_factor"] + w * cal["write_factor"]
        traffic[f"{k}|{WORKLOAD[wl]}"] = t
        rows.append({"pass": tag, "kernel": k, "workload": WORKLOAD[wl], "launches": len(cf[k]["FETCH_SIZE"]), "FETCH_SIZE_KiB": f / 1024, "WRITE_SIZE_KiB": w / 1024,
                     "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": B_ALG[wl], "traffic_over_algorithmic": round(t / B_ALG[wl], 5),
                     "avg_ms_under_pmc": sum(cf[k]["_ms"]) / len(cf[k]["_ms"])})
summ["hbm_traffic"] = rows
# 3. SQ counters
sq = {}
for d in sorted(glob.glob(os.path.join(base, "pmc_sq_*"))):
    if not os.path.isdir(d):
        continue
    for k, cs in counters(d).items():
        if k.startswith("fz_block_kernel"):
            sq[os.path.basename(d)[7:] + ":" + k] = {c: sum(v) / len(v) for c, v in cs.items()}
for k, c in sq.items():
    if "GRBM_GUI_ACTIVE" in c and "_ms" in c:
        c["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (c["_ms"] * 1e6)          # summed over the 8 XCDs
    if "SQ_WAVE_CYCLES" in c:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            c[n + "_share_of_wave_cycles"] = c.get(n, 0) / c["SQ_WAVE_CYCLES"]
summ["sq_counters"] = sq
json.dump(summ, open(os.path.join(outdir, "rocprofv3_summary.json"), "w"), indent=1)
tp = os.path.join(os.path.dirname(outdir.rstrip("/")), "pmc_traffic.json")
t = {"_source": f"{outdir}/rocprofv3_summary.json (tools/profile_round.sh, tools/summarize_round.py); key = '<kernel symbol>|<workload>', "
                "value = HBM bytes per launch from FETCH_SIZE x read_factor + WRITE_SIZE x write_factor (separate --pmc passes)"}
t.update(traffic)
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps(summ["headline_kernel"], indent=1))
for r in rows:
    print(f"{r['pass']:16s} {r['kernel']:36s} traffic/alg {r['traffic_over_algorithmic']:.5f}  {r['avg_ms_under_pmc']:.3f} ms")
print(json.dumps({k: {n: round(x, 4) for n, x in v.items() if 'share' in n or 'clock' in n} for k, v in sq.items()}, indent=1))
