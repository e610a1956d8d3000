#!/usr/bin/env bash
# Dev tool (GPU box): memory-system PMC counters of the headline kernel, tiled vs time-major frames.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_mem; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_REQ_sum TCC_CYCLE_sum TCC_EA0_RDREQ_32B_sum"; do
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
    f=d+"/b_counter_collection.csv"
    if not os.path.exists(f): print("missing", f); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fz_block_kernel_p"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): res[k][tile]=sum(v)/len(v)
for k,v in sorted(res.items()): print(f"{k:46s} tiled8192={v.get('8192',float('nan')):.4g}  timemajor={v.get('0',float('nan')):.4g}")
PY
