#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh batch: HBM traffic and SQ issue share per (kernel CODE id, workload) into profiles/pmc_traffic.json,
profiles/sq_issue_share.json and <profiles/rNN>/rocprofv3_summary.json; the kernel-trace stats of the default bench command next to them.
Entries are {"value", "kernel", "commit", "batch"}: bench.py reports a counter only for the code it was measured on, with where it came from.
usage: tools/merge_profile.py <gpurun_out/dir> <profiles/rNN>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

base, outdir = sys.argv[1:3]
os.makedirs(outdir, exist_ok=True)
prof = os.path.dirname(outdir.rstrip("/"))
commit = open(os.path.join(base, "commit.txt")).read().strip() if os.path.exists(os.path.join(base, "commit.txt")) else "unknown"
batch = os.path.basename(outdir.rstrip("/")) + "/" + os.path.basename(base.rstrip("/"))


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    f = os.path.join(d, "b_counter_collection.csv")
    if not os.path.exists(f):
        return out
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out


def last_json(path):
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path, errors="replace") if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


# calibration on the copy kernel of the headline passes: it reads and writes exactly the frame bytes
head = last_json(os.path.join(base, "pmc_head_FETCH_SIZE.log"))
cf, cw = counters(os.path.join(base, "pmc_head_FETCH_SIZE")), counters(os.path.join(base, "pmc_head_WRITE_SIZE"))
read_factor, write_factor = 2.0, 1.0
copy_bytes = None
if head and "fz_copy_kernel" in cf and "fz_copy_kernel" in cw:
    copy_bytes = head["config"]["streams_per_gpu"] * head["config"]["block_samples"] * 4
    f = sum(cf["fz_copy_kernel"]["FETCH_SIZE"]) / len(cf["fz_copy_kernel"]["FETCH_SIZE"]) * 1024
    w = sum(cw["fz_copy_kernel"]["WRITE_SIZE"]) / len(cw["fz_copy_kernel"]["WRITE_SIZE"]) * 1024
    read_factor, write_factor = copy_bytes / f, copy_bytes / w
summ = {"_source": f"tools/profile_bench.sh -> {base}", "commit": commit, "calibration_on_copy_kernel": {"read_factor": read_factor, "write_factor": write_factor, "copy_bytes": copy_bytes},
        "hbm_traffic": [], "sq_counters": {}}
tp, sp = os.path.join(prof, "pmc_traffic.json"), os.path.join(prof, "sq_issue_share.json")
# (entries of earlier rounds were keyed by kernel symbol: a symbol does not name the code, they are not carried over)
traffic = {k: v for k, v in (json.load(open(tp)) if os.path.exists(tp) else {}).items() if isinstance(v, dict)}
shares = {k: v for k, v in (json.load(open(sp)) if os.path.exists(sp) else {}).items() if isinstance(v, dict)}
shares["_source"] = "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES per '<kernel code id>|<workload>' (tools/profile_bench.sh, tools/merge_profile.py)"
for d in sorted(glob.glob(os.path.join(base, "pmc_*_FETCH_SIZE"))):
    tag = os.path.basename(d)[4:-len("_FETCH_SIZE")]
    log = last_json(d + ".log")
    if not log:
        continue
    if tag == "head":
        obj = {"forced": {"kernel": log["roofline"]["kernel"], "code_id": log["roofline"]["code_id"]}, "workload_key": f"cascade6_{log['config']['streams_per_gpu']}x{log['config']['block_samples']}_timemajor",
               "algorithmic_bytes_per_launch": log["roofline"]["algorithmic_bytes_per_launch"]}
    else:
        obj = next(iter(log.values()))
    plan = obj.get("library_default") or obj.get("forced")
    k, cid, wkey, b_alg = plan["kernel"], plan["code_id"], obj["workload_key"], obj["algorithmic_bytes_per_launch"]
    cf, cw, cs = counters(d), counters(d.replace("FETCH_SIZE", "WRITE_SIZE")), counters(d.replace("FETCH_SIZE", "SQ"))
    # (a block that runs as laps / laps + remainder is several launches of the symbol: traffic per BLOCK = mean x launches per block)
    if k in cf and k in cw:
        per_block = max(1, round(len(cf[k]["FETCH_SIZE"]) / max(1, obj.get("launches_profiled", 0) or len(cf[k]["FETCH_SIZE"]))))
        f = sum(cf[k]["FETCH_SIZE"]) / len(cf[k]["FETCH_SIZE"]) * 1024
        w = sum(cw[k]["WRITE_SIZE"]) / len(cw[k]["WRITE_SIZE"]) * 1024
        # launches per block: algorithmic bytes tell (a lap moves its share)
        t1 = f * read_factor + w * write_factor
        n_per = max(1, round(b_alg / t1)) if t1 > 0 else 1
        t = t1 * n_per
        traffic[f"{cid}|{wkey}"] = {"value": t, "kernel": k, "commit": commit, "batch": batch}
        summ["hbm_traffic"].append({"pass": tag, "kernel": k, "workload": wkey, "launches": len(cf[k]["FETCH_SIZE"]), "code_id": cid, "launches_per_block": n_per,
                                    "FETCH_SIZE_KiB": f / 1024, "WRITE_SIZE_KiB": w / 1024, "traffic_bytes_per_block": t,
                                    "algorithmic_bytes_per_block": b_alg, "traffic_over_algorithmic": round(t / b_alg, 5)})
        print(f"{tag:34s} {k:44s} x{n_per} traffic/alg {t / b_alg:.5f}")
    if k in cs and "SQ_WAVE_CYCLES" in cs[k]:
        c = {n: sum(v) / len(v) for n, v in cs[k].items()}
        if "GRBM_GUI_ACTIVE" in c and c["_ms"] > 0:
            c["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (c["_ms"] * 1e6)          # summed over the 8 XCDs
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            c[n + "_share_of_wave_cycles"] = c.get(n, 0) / c["SQ_WAVE_CYCLES"]
        summ["sq_counters"][f"{tag}:{k}"] = c
        shares[f"{cid}|{wkey}"] = {"value": c["SQ_ACTIVE_INST_ANY_share_of_wave_cycles"], "kernel": k, "commit": commit, "batch": batch}
        print(f"{'':34s} {'':44s} issuing {c['SQ_ACTIVE_INST_ANY_share_of_wave_cycles']:.3f} waiting {c['SQ_WAIT_ANY_share_of_wave_cycles']:.3f} clock {c.get('effective_clock_GHz', 0):.2f} GHz")
traffic["_source"] = "profiles/rNN/rocprofv3_summary.json (tools/profile_bench.sh, tools/merge_profile.py); key = '<kernel code id>|<workload>' (fz_program_kernel_code_id), value = HBM bytes per block from FETCH_SIZE x read_factor + WRITE_SIZE x write_factor (separate --pmc passes)"
json.dump(traffic, open(tp, "w"), indent=1)
json.dump(shares, open(sp, "w"), indent=1)
json.dump(summ, open(os.path.join(outdir, "rocprofv3_summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(base, "trace", "*stats*.csv")):
    shutil.copy(f, os.path.join(outdir, "rocprofv3_" + os.path.basename(f).replace("bench_", "")))
for n, m in (("bench_plain.json", "bench_line_plain.json"), ("bench_trace.json", "bench_line_under_rocprof.json"),
             ("bench_details_plain.json", "bench_details_plain.json"), ("bench_details_trace.json", "bench_details_under_rocprof.json")):
    if os.path.exists(os.path.join(base, n)):
        shutil.copy(os.path.join(base, n), os.path.join(outdir, m))
