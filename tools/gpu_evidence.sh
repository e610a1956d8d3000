#!/usr/bin/env bash
# GPU box: the secondary evidence of a round (everything besides tools/profile_round.sh): full GPU test-suite, the access
# pattern microbenchmark, the stream-major kernels, sweeps at few streams.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-evidence}; mkdir -p $O; cd $R
[ -z "$SKIP_PYTEST" ] && (timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tools/_bin/sm_bench > $O/sm_bench.txt 2>&1
python tools/stream_major_bench.py > $O/stream_major_bench.txt 2>&1
python tools/sm_long_probe.py > $O/sm_long_probe.txt 2>&1
for n in 65536 32768 16384; do
  python tools/sweep.py --graph cascade6 --streams $n --tile 8192 --rounds 40 0,0 1,16,256,8 1,24,256,8 1,16,256,16 > $O/sweep_cascade6_$n.txt 2>&1
done
python tools/sweep.py --graph par4 --streams 65536 --tile 4096 --rounds 20 0,0 1,16 1,32 2,16 > $O/sweep_par4_65536.txt 2>&1
python tools/sweep.py --graph par4f --streams 65536 --tile 8192 --rounds 20 0,0 1,16 1,32 2,16 > $O/sweep_par4f_65536.txt 2>&1
[ -z "$SKIP_PYTEST" ] && tail -4 $O/pytest.log; cat $O/sweep_*.txt | grep -v amdgpu
