#!/bin/bash
# round 3: line-aligned in-run loads of the long-run stream-major body: parity, time, PMC traffic
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r03i; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream_major or in_place or modulators or host_stream" > $O/pytest_sm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sm.txt
tail -4 $O/pytest_sm.txt
python bench.py --only streammajor > $O/sm_leg.json 2> $O/sm_err.txt; cat $O/sm_leg.json | cut -c1-900
python tools/stream_major_bench.py > $O/stream_major.txt 2>&1; head -12 $O/stream_major.txt
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_sm_default_$c -o b -- python $R/bench.py --only streammajor > $O/pmc_sm_default_$c.log 2>&1
done
for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
echo done
