#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_major" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/experiments/exp_r03t.py 2>&1 | grep -v amdgpu > $O/t.txt; cat $O/t.txt
for opts in "-DFZ_DBG_NOLOAD -DFZ_DBG_NOSTORE"; do
  FLOWZ_HIP_EXTRA_OPTS="$opts" timeout 300 python tools/experiments/exp_r03u.py 2>&1 | grep -v amdgpu
done > $O/u.txt 2>&1
cat $O/u.txt
