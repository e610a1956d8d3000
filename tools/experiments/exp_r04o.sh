#!/bin/bash
# GPU box, round 4: the whole GPU suite on the tree with the hold body, the store-merge flag, wide frames in lockstep; then the bench line.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04o; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
python tools/show_bench.py $O/bench_line.json 2>/dev/null | head -80
