#!/bin/bash
# round 3: long-run stream-major body with 64-sample phases at two waves per SIMD
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r03j; mkdir -p $O
export FLOWZ_HIP_NO_PLAN_CACHE=1
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream_major or in_place or modulators or host_stream" > $O/pytest_sm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sm.txt
tail -3 $O/pytest_sm.txt
python tools/stream_major_bench.py > $O/stream_major.txt 2>&1; grep -v amdgpu.ids $O/stream_major.txt
echo done
