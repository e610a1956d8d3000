#!/usr/bin/env python3
"""Merge extra --pmc passes (directories pmc_<tag>_FETCH_SIZE / pmc_<tag>_WRITE_SIZE made like tools/profile_round.sh makes them)
into profiles/pmc_traffic.json and profiles/<round>/rocprofv3_summary.json, with the copy-kernel calibration of that summary.
usage: tools/merge_pmc.py <gpurun_out/dir> <profiles/rNN>"""
import collections
import csv
import glob
import json
import os
import sys

base, outdir = sys.argv[1:3]
WORKLOAD = {"head": "cascade6_1048576x4096_timemajor", "tl": "cascade6_1048576x4096_tile8192", "c2": "cascade6_65536x4096_tile8192", "c2h": "cascade6_32768x4096_tile8192", "c2q": "cascade6_16384x4096_tile8192",
            "c3": "par4_1048576x4096_tile4096", "c3f": "par4f_1048576x4096_timemajor", "c4": "osc6_1048576x4096_tile8192",
            "tm": "cascade6_1048576x4096_timemajor", "sm": "cascade6_1048576x4096_streammajor"}
B_ALG = {"head": 1048576 * (4 * 4096 * 2 + 8 * 14), "tl": 1048576 * (4 * 4096 * 2 + 8 * 14), "c2": 65536 * (4 * 4096 * 2 + 8 * 14), "c2h": 32768 * (4 * 4096 * 2 + 8 * 14), "c2q": 16384 * (4 * 4096 * 2 + 8 * 14),
         "c3": 1048576 * (4 * 4096 * 5 + 8 * 16), "c3f": 1048576 * (4 * 4096 * 2 + 8 * 18), "c4": 1048576 * (4 * 4096 * 2 + 8 * 16 + 4 * 31),
         "tm": 1048576 * (4 * 4096 * 2 + 8 * 14), "sm": 1048576 * (4 * 4096 * 2 + 8 * 14)}


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(d, "b_counter_collection.csv"))):
        out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


tp = os.path.join(os.path.dirname(outdir.rstrip("/")), "pmc_traffic.json")
sp = os.path.join(outdir, "rocprofv3_summary.json")
traffic, summ = json.load(open(tp)), json.load(open(sp))
cal = summ["calibration_on_copy_kernel"]
for d in sorted(glob.glob(os.path.join(base, "pmc_*_FETCH_SIZE"))):
    tag = os.path.basename(d)[4:-len("_FETCH_SIZE")]
    wl = tag.split("_")[0]
    if wl not in WORKLOAD or not os.path.exists(os.path.join(d, "b_counter_collection.csv")):
        continue
    cf, cw = counters(d), counters(d.replace("FETCH_SIZE", "WRITE_SIZE"))
    for k in cf:
        if not k.startswith("fz_block_kernel") or k not in cw:
            continue
        f = sum(cf[k]["FETCH_SIZE"]) / len(cf[k]["FETCH_SIZE"]) * 1024
        w = sum(cw[k]["WRITE_SIZE"]) / len(cw[k]["WRITE_SIZE"]) * 1024
        t = f * cal["read_factor"] + w * cal["write_factor"]
        traffic[f"{k}|{WORKLOAD[wl]}"] = t
        summ["hbm_traffic"] = [r for r in summ["hbm_traffic"] if not (r["kernel"] == k and r["workload"] == WORKLOAD[wl])]
        summ["hbm_traffic"].append({"pass": tag + " (separate batch, same calibration)", "kernel": k, "workload": WORKLOAD[wl],
                                    "launches": len(cf[k]["FETCH_SIZE"]), "FETCH_SIZE_KiB": f / 1024, "WRITE_SIZE_KiB": w / 1024, "traffic_bytes_per_launch": t,
                                    "algorithmic_bytes_per_launch": B_ALG[wl], "traffic_over_algorithmic": round(t / B_ALG[wl], 5)})
        print(f"{tag:16s} {k:40s} traffic/alg {t / B_ALG[wl]:.5f}")
# SQ counter passes of the same batch (directories pmc_sq_<object>, as tools/profile_r03.sh makes them)
for d in sorted(glob.glob(os.path.join(base, "pmc_sq_*"))):
    f = os.path.join(d, "b_counter_collection.csv")
    if not (os.path.isdir(d) and os.path.exists(f)):
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, cs in per.items():
        if not k.startswith("fz_block_kernel"):
            continue
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        if "GRBM_GUI_ACTIVE" in c:
            c["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (c["_ms"] * 1e6)          # summed over the 8 XCDs
        if "SQ_WAVE_CYCLES" in c:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                c[n + "_share_of_wave_cycles"] = c.get(n, 0) / c["SQ_WAVE_CYCLES"]
        summ.setdefault("sq_counters", {})[os.path.basename(d)[7:] + ":" + k] = c
        print(f"sq {os.path.basename(d)[7:]:12s} {k:40s} " + "  ".join(f"{n.replace('_share_of_wave_cycles', '')} {x:.3f}" for n, x in c.items() if "share" in n or "clock" in n))
json.dump(traffic, open(tp, "w"), indent=1)
json.dump(summ, open(sp, "w"), indent=1)
