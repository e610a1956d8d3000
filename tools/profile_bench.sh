#!/usr/bin/env bash
# GPU box: the rocprofv3 evidence of the bench line (round 5: the counter tables are keyed by the kernels' CODE ids).
#   usage: gpurun -- "FZ_COMMIT=$(git rev-parse --short HEAD) tools/profile_bench.sh <name>"   -> gpurun_out/<name>/...
#   then here: tools/merge_profile.py gpurun_out/<name> profiles/r06      (refreshes profiles/pmc_traffic.json, sq_issue_share.json)
# 1. the DEFAULT bench command plain, then under --kernel-trace --stats (what the driver runs);
# 2. per workload x layout of the line (bench.py --only <spec>:<layout>[:tile], the library's static choice): separate --pmc passes --
#    never combined with tracing -- of FETCH_SIZE, WRITE_SIZE (HBM traffic) and of the SQ counters (issue share, real clock).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof_r06}; mkdir -p $O; echo "${FZ_COMMIT:-unknown}" > $O/commit.txt; cd /tmp; export TMPDIR=/tmp
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
if [ -z "${PASSES_ONLY:-}" ]; then
BENCH_DETAILS=$O/bench_details_plain.json python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
BENCH_DETAILS=$O/bench_details_trace.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_trace.json 2> $O/bench_trace.err
fi
HEAD="--steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-config34 --no-sustained --no-layout-legs --no-extras --no-next-rows --no-autotune"
pass() {   # pass <tag> <bench args...>: one run per counter set   (ONLY_TAGS=<regex>: just the matching passes)
  tag=$1; shift
  if [ -n "${ONLY_TAGS:-}" ] && ! echo "$tag" | grep -Eq "$ONLY_TAGS"; then return; fi
  export BENCH_DETAILS=/tmp/bench_details_pass.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_FETCH_SIZE -o b -- python $R/bench.py "$@" > $O/pmc_${tag}_FETCH_SIZE.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${tag}_WRITE_SIZE -o b -- python $R/bench.py "$@" > $O/pmc_${tag}_WRITE_SIZE.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_${tag}_SQ -o b -- python $R/bench.py "$@" > $O/pmc_${tag}_SQ.log 2>&1
}
pass head $HEAD
L="--no-autotune --reps-ms 5 --only"
for leg in cascade6_1048576:tiled:8192 cascade6_1048576:stream_major cascade6_262144:stream_major \
           cascade6_65536:tiled:8192 cascade6_65536:time_major cascade6_65536:stream_major cascade6_32768:tiled:8192 cascade6_16384:tiled:8192 \
           par4:tiled:4096 par4:time_major par4:stream_major par4f:time_major osc6:tiled:8192 osc6:time_major osc6:stream_major \
           cascade6_1000000:time_major cascade6_1048577:time_major cascade6_786432:time_major cascade6_2097152:time_major \
           df1:time_major df2:time_major df1t:time_major df2t:time_major \
           lds_ring:time_major far_ring:time_major blocks64:time_major modulated:time_major double_biquad:time_major complex_one_pole:time_major; do
  pass $(echo $leg | tr ':' '-') $L $leg
done
# rocprofv3 nests its output under <hostname>/: flatten; the per-launch trace is tens of MiB: the stats are what is kept
for d in $O/trace $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
rm -f $O/trace/*kernel_trace.csv
ls $O | wc -l
tail -c 400 $O/bench_plain.json
