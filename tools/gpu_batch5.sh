#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-batch5}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "typed or stream_major" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tools/_bin/sm_bench > $O/sm_bench.txt 2>&1
python tools/stream_major_bench.py > $O/stream_major_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -o b -- python $R/bench.py --only config2 > $O/pmc_sq1.log 2>&1
for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
python - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$O/pmc_sq*")):
    f = d + "/b_counter_collection.csv"
    if not os.path.exists(f): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fz_block_kernel"):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[r["Kernel_Name"]]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in agg.items():
        for c, v in sorted(cs.items()): print(f"{k:40s} {c:26s} mean {sum(v)/len(v):.6g}  n={len(v)}")
PY
tail -15 $O/pytest.log | cut -c1-300; cat $O/stream_major_bench.txt | grep -v adapter; tail -12 $O/sm_bench.txt
