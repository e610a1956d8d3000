#!/usr/bin/env bash
# Dev tool (GPU box): one exploratory batch -- GPU test-suite, microbenchmarks, tune logs, SQ counters of config 2.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-batch}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tools/_bin/sm_bench > $O/sm_bench.txt 2>&1
python tools/stream_major_bench.py > $O/stream_major_bench.txt 2>&1
FLOWZ_HIP_TUNE_LOG=1 python bench.py --no-cpu-baseline --no-sustained > $O/bench_tunelog.json 2> $O/tune_log.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -o b -- python $R/bench.py --only config2 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o b -- python $R/bench.py --only config2 > $O/pmc_sq2.log 2>&1
for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_sq*")):
    import os
    f = d + "/b_counter_collection.csv"
    if not os.path.exists(f): print("missing", f); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fz_block_kernel"):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in sorted(cs.items()): print(f"{k:40s} {c:26s} mean {sum(v)/len(v):.6g}  n={len(v)}")
PY
tail -3 $O/pytest.log
