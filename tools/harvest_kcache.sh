#!/bin/bash
# GPU box: the GPU suite with per-test durations; the code objects the tests had to JIT on the box are harvested into gpurun_out/<dir>/kcache so that
# the next snapshot carries them in zignal_amd/_kcache (the cache is content-addressed: source + options + compiler; a kernel source that changed
# simply misses and is built again).  Round 4: 784 GPU tests take 550-785 s on a box that JITs ~1800 kernels through the compiler worker, 80 s with them cached.
#   usage (here):  gpurun -- 'bash tools/harvest_kcache.sh [dir]'  &&  cp -n gpurun_out/<dir>/kcache/*.hsaco zignal_amd/_kcache/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-harvest}; mkdir -p $O/kcache
touch /tmp/fz_marker; sleep 1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -75 $O/pytest_gpu.txt | cut -c1-160
find zignal_amd/_kcache -newer /tmp/fz_marker -type f \( -name '*.hsaco' -o -name '*.txt' -o -name '*.json' \) ! -name 'plans.txt' -exec cp {} $O/kcache/ \;
ls $O/kcache | wc -l; du -sh $O/kcache
