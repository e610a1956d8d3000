#!/bin/bash
# round 3, last session: the PAIR long-run stream-major body -- its GPU tests, the bench leg, PMC traffic of the new default kernel
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03pair; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_major" > $O/pytest_stream_major.txt 2>&1; tail -3 $O/pytest_stream_major.txt
timeout 300 python bench.py --only streammajor 2> /dev/null > $O/bench_streammajor.json; python - <<'P'
import json
d = json.load(open("gpurun_out/r03pair/bench_streammajor.json"))["streammajor"]
for k in ("library_default", "tuned"):
    print(k, d[k]["kernel"], d[k]["avg_launch_ms"], d[k]["frac"])
print(d["candidates_ms"], d["parity"])
P
R=$PWD; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_sm_$c -o b -- python $R/bench.py --only streammajor --no-autotune > $R/$O/pmc_sm_$c.log 2>&1
done
cd $R; for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
ls $O
