#!/bin/bash
# GPU box, round 4: the cross-paired wave split again (parity after the stash fix), and what bounds its rounds: the same kernel without
# memory traffic of the I/O wave (-DFZ_DBG_NOIO), without barriers (-DFZ_DBG_NOBARRIER), without arithmetic (-DFZ_DBG_NOCOMPUTE).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04f; mkdir -p $O
X=16812032
export FLOWZ_HIP_AUTOTUNE=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "cross" > $O/pytest_cross.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_cross.txt
tail -5 $O/pytest_cross.txt
S="timeout 600 python tools/sweep.py --rounds 9"
for opts in "" "-DFZ_DBG_NOIO" "-DFZ_DBG_NOBARRIER" "-DFZ_DBG_NOCOMPUTE" "-DFZ_DBG_NOCOMPUTE -DFZ_DBG_NOIO" "-DFZ_DBG_NOIO -DFZ_DBG_NOBARRIER"; do
  echo "## EXTRA_OPTS=[$opts]"
  FLOWZ_HIP_EXTRA_OPTS="$opts" $S --streams 16384 --tile 8192 1,16,64,$X 1,32,64,$X 1,32,64,34816 2>&1 | grep -v amdgpu.ids
  FLOWZ_HIP_EXTRA_OPTS="$opts" $S --streams 32768 --tile 8192 1,16,64,$X 1,16,128,$X 2>&1 | grep -v amdgpu.ids
done > $O/sweeps.txt 2>&1
cat $O/sweeps.txt
