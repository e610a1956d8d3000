#!/usr/bin/env bash
# GPU box: the rocprofv3 evidence of round 3's bench line.   usage: tools/profile_r03.sh <name>  -> gpurun_out/<name>/...
# 1. the default bench command plain and under --kernel-trace --stats (what the driver runs);
# 2. separate --pmc passes (never combined with tracing) of FETCH_SIZE and WRITE_SIZE, one pair per bench object: the first big
#    launch of every shape measures the candidates at hand, so ONE pass sees the library default and every plan it may pick;
# 3. SQ counters (issue / wait split, effective clock) of the headline, the two layout legs and the few-stream objects.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof03}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_trace.log 2>&1
HEAD="--steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-config34 --no-sustained --no-layout-legs"
pass() {   # pass <tag> <bench args...>: one FETCH_SIZE and one WRITE_SIZE run
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${tag}_$c -o b -- python $R/bench.py "$@" > $O/pmc_${tag}_$c.log 2>&1
  done
}
pass head_default $HEAD
pass tl_default   --only tiled
pass sm_default   --only streammajor
pass c2_default   --only config2
pass c2h_default  --only config2h
pass c2q_default  --only config2q
pass c3_default   --only config3
pass c3f_default  --only config3f
pass c4_default   --only config4
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_head -o b -- python $R/bench.py $HEAD > $O/pmc_sq_head.log 2>&1
for o in tiled streammajor config2 config2h config2q; do
  rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_$o -o b -- python $R/bench.py --only $o --no-autotune > $O/pmc_sq_$o.log 2>&1
done
for d in $O/trace $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
ls $O | wc -l
tail -1 $O/bench_plain.log | cut -c1-400
