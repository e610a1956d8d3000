#!/bin/bash
# Round 5: the period of the XCD-wide step (FZ_GS_PERIOD, chunks between arrivals) of the lockstep walks, per workload.  PREBUILD=1 first.
cd "$(dirname "$0")/../.."
PB=${PREBUILD:+--prebuild}
L=$((524288+8388608))
for per in 0 2 4 8 16 32; do
  [ $per = 0 ] && unset FLOWZ_HIP_EXTRA_OPTS || export FLOWZ_HIP_EXTRA_OPTS="-DFZ_GS_PERIOD_OVERRIDE=$per"
  echo "## period $per (0 = default)"
  python tools/sweep.py $PB --graph cascade6 --streams 1048576 --reps 4 --rounds 3 4,1,1024,$((L+32)) 2>&1 | grep -v "amdgpu\|^#"
  python tools/sweep.py $PB --graph par4 --streams 1048576 --reps 2 --rounds 3 1,3,1024,$L 2>&1 | grep -v "amdgpu\|^#"
  python tools/sweep.py $PB --graph c32onepole --streams 1048576 --reps 3 --rounds 3 4,1,1024,$((L+32)) 2>&1 | grep -v "amdgpu\|^#"
  python tools/sweep.py $PB --graph osc --streams 1048576 --reps 4 --rounds 3 1,4,1024,$((L+8)) 2>&1 | grep -v "amdgpu\|^#"
  python tools/sweep.py $PB --graph cascade6 --streams 2097152 --reps 2 --rounds 3 2,2,1024,$L 2>&1 | grep -v "amdgpu\|^#"
done
