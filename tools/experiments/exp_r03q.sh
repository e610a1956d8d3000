#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03q; mkdir -p $O
timeout 1200 python -m pytest tests/test_bench_contract.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
python tools/show_bench.py $O/bench_line.json | cut -c1-330
