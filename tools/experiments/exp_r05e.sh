#!/bin/bash
# Round 5 (GPU box): address-translation and L2 counters of ONE lap of the 4-wire sum on dense rows (262 144 streams) and on strided rows (a quarter of rows
# 16 MiB apart: lap 1 of 1 048 576 streams), the same kernel (one stream per lane, two-row chunks, 1024 lanes, lockstep + XCD step)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05e}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export FZ_VARIANT=1,2,1024,8912896
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_TAG_STALL_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/dense_$tag -o b -- python $R/tools/experiments/exp_r05d.py 262144 > $O/dense_$tag.log 2>&1
  FLOWZ_HIP_ONLY_LAP=1 rocprofv3 --pmc $set --output-format csv -d $O/strided_$tag -o b -- python $R/tools/experiments/exp_r05d.py 1048576 > $O/strided_$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, os
O="$O"
for d in sorted(glob.glob(O+"/*_*")):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+"/**/b_counter_collection.csv", recursive=True)
    if not fs: print("missing", d, open(d+".log").read()[-300:]); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith("fz_block_kernel_p"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(os.path.basename(d), {k: round(sum(v)/len(v), 3) for k,v in agg.items()}, "launches", len(agg["_ms"]))
PY
