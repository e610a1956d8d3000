#!/bin/bash
# GPU box, round 4: the default bench command plain and under rocprofv3 --kernel-trace --stats once more, with every tune candidate of the final sources in
# the kernel cache (the batch of tools/profile_r04.sh before it ran on a cache that build() had not refilled: the first-launch plan measurement only
# takes kernels at hand); then exp_r04u.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04v; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py > $O/bench_trace.json 2> $O/bench_trace.err
for d in $O/trace; do [ -d "$d" ] && find $d -mindepth 2 -name '*.csv' -exec mv {} $d/ \; ; done
rm -f $O/trace/*kernel_trace.csv
cd $R; python tools/show_bench.py $O/bench_plain.json | cut -c1-200
bash tools/experiments/exp_r04u.sh
