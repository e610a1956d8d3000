#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-batch4}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
FLOWZ_HIP_TUNE_LOG=1 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/tune_log.txt
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 1,20,256,8 1,16,256,40 1,24,256,40 > $O/sweep_config2.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 > $O/sweep_32k.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -o b -- python $R/bench.py --only config2 > $O/pmc_sq1.log 2>&1
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
    for k, cs in agg.items():
        for c, v in sorted(cs.items()): print(f"{k:40s} {c:26s} mean {sum(v)/len(v):.6g}  n={len(v)}")
PY
tail -15 $O/pytest.log; cat $O/sweep_config2.txt $O/sweep_32k.txt
