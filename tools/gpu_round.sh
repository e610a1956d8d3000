#!/bin/bash
# GPU box: the -m gpu test suite (recording the kernel manifest and harvesting the code objects it had to JIT), then the default bench line.
#   usage: gpurun --timeout 2400 -- 'bash tools/gpu_round.sh <name> [pytest args]'   -> gpurun_out/<name>/
#   afterwards here: cp -n gpurun_out/<name>/kcache/*.hsaco zignal_amd/_kcache/ ; gzip -9c gpurun_out/<name>/manifest.fzm > tests/kernel_manifest.fzm.gz
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r05}; mkdir -p $O/kcache; shift
touch /tmp/fz_marker; sleep 1
rm -f $O/manifest.fzm
FLOWZ_HIP_MANIFEST=$PWD/$O/manifest.fzm timeout 2000 python -m pytest tests -m gpu -q --durations=30 "$@" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -45 $O/pytest_gpu.txt | cut -c1-200
BENCH_DETAILS=$O/bench_details.json FLOWZ_HIP_MANIFEST=$PWD/$O/manifest.fzm timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
tail -c 600 $O/bench_err.txt
find zignal_amd/_kcache -newer /tmp/fz_marker -type f -name '*.hsaco' -exec cp {} $O/kcache/ \;
ls $O/kcache | wc -l; du -sh $O/kcache; wc -c $O/manifest.fzm
wc -c $O/bench_line.json; cat $O/bench_line.json
