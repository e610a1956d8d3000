#!/bin/bash
# GPU box: the round's last verification -- the -m gpu suite, the default bench line plain and under rocprofv3 --kernel-trace --stats
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/${1:-final}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/trace.log; echo "trace rc=$?"
cd $R; find $O/trace -mindepth 2 -name '*.csv' -exec mv {} $O/trace/ \; 2>/dev/null
ls $O/trace | head; rm -f $O/trace/*kernel_trace.csv                     # (the per-launch trace is tens of MiB: the stats are what is kept)
python tools/show_bench.py $O/bench_line.json 2>/dev/null | head -12
