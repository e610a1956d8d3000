#!/bin/bash
# GPU box: the -m gpu test suite, then the default bench line (outputs under gpurun_out/$1)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
tail -c 600 $O/bench_err.txt
python tools/show_bench.py $O/bench_line.json 2>/dev/null | head -60
