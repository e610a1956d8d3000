#!/usr/bin/env bash
# GPU box: the rocprofv3 evidence of a round's bench line.
#   usage: tools/profile_round.sh <name>      -> gpurun_out/<name>/...   (summarise with tools/summarize_round.py)
# 1. kernel trace + stats of the DEFAULT bench command (what the driver runs);
# 2. separate --pmc passes (never combined with tracing) of FETCH_SIZE and WRITE_SIZE for every BASELINE config, the
#    library-default variant and the variants fz_program_tune picks on the boards of this pool.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ -z "$PASSES_ONLY" ]; then
python $R/bench.py > $O/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_trace.log 2>&1
fi
HEAD="--steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-config34 --no-sustained"
pass() {   # pass <tag> <bench args...>: one FETCH_SIZE and one WRITE_SIZE run   (ONLY_TAGS=<regex>: just the matching passes)
  tag=$1; shift
  if [ -n "$ONLY_TAGS" ] && ! echo "$tag" | grep -Eq "$ONLY_TAGS"; then return; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${tag}_$c -o b -- python $R/bench.py "$@" > $O/pmc_${tag}_$c.log 2>&1
  done
}
pass head_default  $HEAD --no-autotune
pass head_p4u4     $HEAD --lanes 4 --unroll 4 --block 256 --flags 2097152
pass head_p4u8b128 $HEAD --lanes 4 --unroll 8 --block 128 --flags 2097152
pass head_p2u32wg2 $HEAD --lanes 2 --unroll 32 --block 256 --flags 2097152
pass head_p2u16    $HEAD --lanes 2 --unroll 16
pass head_p4u8     $HEAD --lanes 4 --unroll 8
pass head_p2u16wg2 $HEAD --lanes 2 --unroll 16 --block 256 --flags 2097152
pass head_p4u8wg1  $HEAD --lanes 4 --unroll 8 --block 256 --flags 1048576
pass head_p4u12wg1 $HEAD --lanes 4 --unroll 12 --block 256 --flags 1048576
pass head_p4u16wg1 $HEAD --lanes 4 --unroll 16 --block 256 --flags 1048576
pass c2_default    --only config2
pass c2_u24        --only config2 --lanes 1 --unroll 24 --flags 8
pass c2_io         --only config2 --lanes 1 --unroll 16 --flags 32768
pass c2h_default   --only config2h
pass c2h_io        --only config2h --lanes 1 --flags 33792
pass c2h_u32       --only config2h --lanes 1 --unroll 32 --flags 1024
pass c2q_default   --only config2q
pass c2q_io        --only config2q --lanes 1 --flags 34816
pass c3_default    --only config3
pass c3_u16wg1     --only config3 --lanes 1 --unroll 16 --block 256 --flags 1048576
pass c3_u8wg1      --only config3 --lanes 1 --unroll 8 --block 256 --flags 1048576
pass c3f_default   --only config3f
pass c3f_p4u8b128  --only config3f --lanes 4 --unroll 8 --block 128 --flags 2097152
pass c3f_p4u8wg1   --only config3f --lanes 4 --unroll 8 --block 256 --flags 1048576
# every other plan fz_program_tune may select for the fan-out sum and the oscillator chain
for cfg in c3f:config3f c4:config4; do
  t=${cfg%%:*}; o=${cfg##*:}
  pass ${t}_p2u16     --only $o --lanes 2 --unroll 16
  pass ${t}_p4u8      --only $o --lanes 4 --unroll 8
  pass ${t}_p2u16wg2  --only $o --lanes 2 --unroll 16 --block 256 --flags 2097152
  pass ${t}_p2u32wg2  --only $o --lanes 2 --unroll 32 --block 256 --flags 2097152
  pass ${t}_p4u12wg1  --only $o --lanes 4 --unroll 12 --block 256 --flags 1048576
  pass ${t}_p4u16wg1  --only $o --lanes 4 --unroll 16 --block 256 --flags 1048576
  pass ${t}_p4u4wg2   --only $o --lanes 4 --unroll 4 --block 256 --flags 2097152
done
pass c4_p4u8wg1    --only config4 --lanes 4 --unroll 8 --block 256 --flags 1048576
pass c4_default    --only config4
pass c4_p4u8b128   --only config4 --lanes 4 --unroll 8 --block 128 --flags 2097152
[ -n "$PASSES_ONLY" ] && { for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done; ls $O | wc -l; exit 0; }
# SQ counters of the headline default and of config 2 (issue / wait split, real clock)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_head -o b -- python $R/bench.py $HEAD --no-autotune > $O/pmc_sq_head.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c2 -o b -- python $R/bench.py --only config2 > $O/pmc_sq_c2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c2h -o b -- python $R/bench.py --only config2h > $O/pmc_sq_c2h.log 2>&1
# rocprofv3 nests its output under <hostname>/: flatten
for d in $O/trace $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
ls $O | head -80
tail -1 $O/bench_plain.log | cut -c1-600
