#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-batch3}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
python tools/sweep.py --graph cascade6 --streams 65536 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 1,32,256,8 1,8,256,8 1,12,256,8 1,16,256,40 1,24,256,40 > $O/sweep_config2.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 32768 --tile 8192 --rounds 40 1,16,256,8 1,24,256,8 1,32,256,8 1,16,256,40 > $O/sweep_32k.txt 2>&1
python tools/sweep.py --graph cascade6 --streams 16384 --tile 8192 --rounds 40 1,16,256,8 1,32,256,8 1,16,128,8 1,16,64,8 > $O/sweep_16k.txt 2>&1
FLOWZ_HIP_TUNE_LOG=1 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/tune_log.txt
python tools/stream_major_bench.py > $O/stream_major_bench.txt 2>&1
tail -3 $O/pytest.log; cat $O/sweep_config2.txt $O/sweep_32k.txt $O/sweep_16k.txt; cat $O/stream_major_bench.txt | grep -v adapter
