#!/usr/bin/env python3
"""Per time step of one wave's 64 streams: the SQ counters of the config-2 arrangements (passes of tools/experiments/exp_r06a.sh).
usage: exp_r06_floor_table.py <gpurun_out/r06a>"""
import collections
import csv
import glob
import json
import os
import sys

base = sys.argv[1]
NS, T = 65536, 4096
STEPS = NS // 64 * T                       # (64 streams, one time step): the unit every arrangement is priced in


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out


for tag in ("packed", "w2", "io2", "w3"):
    row = {"arrangement": tag}
    for p in "AB":
        c = counters(os.path.join(base, f"pmc_{tag}_{p}"))
        ks = [k for k in c if k.startswith("fz_block_kernel")]
        if not ks:
            continue
        k = max(ks, key=lambda q: len(c[q]["_ms"]))
        row["kernel"] = k
        for name, vals in c[k].items():
            m = sum(vals) / len(vals)
            if name == "_ms":
                row[f"ms_pass{p}"] = round(m, 4)
            elif name == "SQ_WAVES":
                row["waves"] = m
            elif name == "GRBM_GUI_ACTIVE":
                row[f"sclk_MHz_pass{p}"] = round(m / (sum(c[k]["_ms"]) / len(c[k]["_ms"])) / 1e3, 0)
            elif name in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
                row[name + "_x4_per_step"] = round(4 * m / STEPS, 2)        # (these count quad-cycles)
            else:
                row[name + "_per_step"] = round(m / STEPS, 3)
    print(json.dumps(row))
