#!/usr/bin/env bash
# GPU box: where does the time of a lone wave per SIMD go?  The config-2 kernel (6-biquad cascade, stage-packed, one stream per
# lane) at 65536 and 32768 streams with the main loop's loads and / or stores sent through zero-byte descriptors (same
# instruction stream, no memory traffic): what is left is issue time + dependency stalls.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-probe}; mkdir -p $O; cd $R
for n in 65536 32768; do
  for opts in "" "-DFZ_DBG_NOLOAD" "-DFZ_DBG_NOSTORE" "-DFZ_DBG_NOLOAD -DFZ_DBG_NOSTORE"; do
    echo "## streams=$n opts='$opts'"
    FLOWZ_HIP_EXTRA_OPTS="$opts" python tools/sweep.py --graph cascade6 --streams $n --tile 8192 --rounds 30 1,16,256,8 1,8,256,8 1,32,256,8 2>&1 | grep -v amdgpu
  done
done > $O/wave_time_probe.txt 2>&1
cat $O/wave_time_probe.txt
