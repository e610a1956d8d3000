#!/usr/bin/env bash
# Round 6, batch g:   gpurun --timeout 900 -- 'bash tools/experiments/exp_r06g.sh'   -> gpurun_out/r06g/
#  the C++ front end's tests on hardware (flowz/shard.hpp: the statistics reduction over RCCL from C++), then one more default bench line (another board for the ranges)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_cpp_edsl.py tests/test_c_abi.py -m gpu -q > $O/pytest_cpp.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_cpp.txt; tail -15 $O/pytest_cpp.txt | cut -c1-250
tests/cpp/_build/test_two_devices_gpu 2>&1 | tail -4 | tee $O/two_devices.txt
BENCH_DETAILS=$O/bench_details.json timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"; cut -c1-600 $O/bench_line.json
