#!/bin/bash
# GPU box, round 4: the GPU suite with per-test durations; the code objects the tests had to JIT on the box are harvested into gpurun_out/ so that
# the next snapshot carries them in zignal_amd/_kcache (the cache is content-addressed: source + options + compiler).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04q; mkdir -p $O/kcache
touch /tmp/fz_marker; sleep 1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -75 $O/pytest_gpu.txt | cut -c1-160
find zignal_amd/_kcache -newer /tmp/fz_marker -type f \( -name '*.hsaco' -o -name '*.txt' -o -name '*.json' \) ! -name 'plans.txt' -exec cp {} $O/kcache/ \;
ls $O/kcache | wc -l; du -sh $O/kcache
