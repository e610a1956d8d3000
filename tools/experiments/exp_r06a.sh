#!/usr/bin/env bash
# Round 6, first GPU batch:   gpurun --timeout 2700 -- 'bash tools/experiments/exp_r06a.sh'   -> gpurun_out/r06a/
#  0. the GPU tests this round changed (reference-misrouted graph, RBJ generator bit parity, opt-in autotune)
#  1. config 2 (65 536 x 4096): the wave split W = 2 with both waves of a tuple on the SAME SIMD (block 256 = four tuples per workgroup) next to the
#     stage-packed single wave and the two I/O waves, plain rows and tiles; board power / clock / joules per launch; SQ counters per arrangement
#  2. placement: LDS rings, oscillator chain, 64-sample windows on six fresh allocations, free-running default against lockstep geometries
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
export FLOWZ_HIP_NO_PLAN_CACHE=1
timeout 900 python -m pytest tests -m gpu -q -x -k "misroutes or rbj or autotune or c_program or tuned_plan" > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.txt
tail -5 $O/pytest_subset.txt
S="timeout 600 python tools/sweep.py --graph cascade6 --streams 65536 --rounds 7 --reps 50"
VARS="0,0 1,16,256,1024 1,32,256,1024 1,8,256,1024 1,16,128,1024 1,16,512,1024 1,16,0,33587200 1,16,256,2048 1,16,0,32768 1,24,256,8"
$S $VARS > $O/config2_time_major.txt 2>&1
$S --tile 8192 $VARS > $O/config2_tiled.txt 2>&1
$S --sm 0,0 1,128,0,256 1,128,64,256 1,64,0,256 > $O/config2_stream_major.txt 2>&1
cat $O/config2_time_major.txt | cut -c1-160
timeout 600 python tools/experiments/exp_r06_config2_power.py > $O/config2_power.txt 2>&1
cat $O/config2_power.txt | cut -c1-220
cd /tmp; export TMPDIR=/tmp
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SQB="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_LDS"
SQC="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
B="python $R/bench.py --no-autotune --reps-ms 8 --only cascade6_65536:time_major"
pmc() {  # pmc <tag> <variant args...>
  tag=$1; shift
  export BENCH_DETAILS=/tmp/bench_details_pass.json
  timeout 300 rocprofv3 --pmc $SQA --output-format csv -d $O/pmc_${tag}_A -o b -- $B "$@" > $O/pmc_${tag}_A.log 2>&1
  timeout 300 rocprofv3 --pmc $SQB --output-format csv -d $O/pmc_${tag}_B -o b -- $B "$@" > $O/pmc_${tag}_B.log 2>&1
  if ! find $O/pmc_${tag}_B -name '*counter_collection.csv' | grep -q .; then
    timeout 300 rocprofv3 --pmc $SQC --output-format csv -d $O/pmc_${tag}_B -o b -- $B "$@" > $O/pmc_${tag}_B.log 2>&1
  fi
}
pmc packed
pmc w2 --lanes 1 --unroll 16 --block 256 --flags 1024
pmc io2 --lanes 1 --unroll 16 --flags 33587200
pmc w3 --lanes 1 --unroll 16 --block 256 --flags 2048
for d in $O/pmc_*; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
cd $R
python tools/experiments/exp_r06_floor_table.py $O > $O/config2_floor_counters.txt 2>&1; cat $O/config2_floor_counters.txt | cut -c1-250
for g in ldsring osc blocks64; do timeout 900 python tools/experiments/exp_r06_placement.py $g > $O/placement_$g.txt 2>&1; cat $O/placement_$g.txt | cut -c1-400; done
du -sh $O
