#!/usr/bin/env bash
# Dev tool (GPU box): address-translation counters of the headline kernel, tiled vs time-major frames.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_tlb; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_PERMISSION_MISS_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  for tile in 8192 0; do
    rocprofv3 --pmc $set --output-format csv -d $O/s${i}_t${tile} -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-autotune --no-config2 --tile $tile > $O/s${i}_t${tile}.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, os
O="$O"
res=collections.defaultdict(dict)
for d in sorted(glob.glob(O+"/s*_t*")):
    if not os.path.isdir(d): continue
    tile=d.split("_t")[-1]
    fs=glob.glob(d+"/**/b_counter_collection.csv", recursive=True)
    if not fs: print("missing", d, open(d+".log").read()[-300:]); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith("fz_block_kernel_p"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): res[k][tile]=sum(v)/len(v)
for k,v in sorted(res.items()): print(f"{k:40s} tiled8192={v.get('8192',float('nan')):.5g}  timemajor={v.get('0',float('nan')):.5g}")
PY
