#!/bin/bash
# GPU box, round 4: the whole -m gpu suite (all failures, not the first), the bench line, then: 2 M streams geometries; where the walk in
# lockstep starts to pay over the block length; stream-major workgroup sizes for the smaller shapes.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04d; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896; LGP=8912928
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
sweeps() {
export FLOWZ_HIP_AUTOTUNE=0
$S --streams 2097152 2,2,1024,$LG 2,1,1024,$LGP 4,2,512,$LG 4,1,1024,$LGP 2,4,512,$LG
for T in 64 128 256 512 1024; do $S --graph params6 --samples $T 2,1,1024,$LGP 2,16,256,0 1,16,256,0; done
for T in 128 256 512; do $S --samples $T 4,1,1024,$LGP 2,16,256,0; done
$S --streams 1048577 0,0,0,0
$S --streams 1114112 0,0,0,0
$S --streams 1000001 0,0,0,0
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"
tail -c 600 $O/bench_err.txt
python tools/show_bench.py $O/bench_line.json > $O/bench_summary.txt 2>&1; cat $O/bench_summary.txt
sweeps > $O/sweeps.txt 2>&1; grep -v amdgpu.ids $O/sweeps.txt
