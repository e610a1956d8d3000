#!/usr/bin/env bash
# Dev tool (GPU box): the rocprofv3 evidence of the headline bench line.
#   usage: tools/profile_headline.sh <name>      -> gpurun_out/<name>/{trace, pmc_*, *.log}
# Kernel trace + stats of the DEFAULT bench command, then separate --pmc passes (never combined
# with tracing); summarise with tools/summarize_prof.py into profiles/rNN/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_trace.log 2>&1
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-autotune --no-config2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -o bench -- python $R/bench.py $PMC_ARGS > $O/pmc_FETCH_SIZE.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -o bench -- python $R/bench.py $PMC_ARGS > $O/pmc_WRITE_SIZE.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o bench -- python $R/bench.py $PMC_ARGS > $O/pmc_sq.log 2>&1
# rocprofv3 nests its output under <hostname>/: flatten
for d in $O/trace $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
ls $O $O/trace
tail -1 $O/bench_plain.log | cut -c1-400
