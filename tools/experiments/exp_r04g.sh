#!/bin/bash
# GPU box, round 4: two I/O waves per tuple (FZ_VF_IO_WAVE2: a loader and a storer) for the wave-split kernels: parity, then few-stream sweeps.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04g; mkdir -p $O
MODE=${1:-run}
X=16812032; X2=50366464; W3=34816; W32=33589248; W22=33588224; W2=33792; W12=33587200; W1=32768
export FLOWZ_HIP_AUTOTUNE=0
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 9"; fi
sweeps() {
$S --streams 16384 --tile 8192 1,32,64,$W3 1,32,64,$W32 1,16,64,$W32 1,32,64,$X 1,32,64,$X2 1,16,64,$X2
$S --streams 32768 --tile 8192 1,32,128,$W2 1,32,128,$W22 1,16,128,$W22 1,32,64,$W32 1,16,64,$W32 1,16,64,$X2 1,16,128,$X2
$S --streams 65536 --tile 8192 1,16,256,8 1,16,256,$W1 1,16,256,$W12 1,16,128,$W22 1,16,64,$W32 1,8,128,$X2
$S --streams 8192 --tile 8192 1,32,64,$W3 1,32,64,$W32 1,32,64,$X2
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "cross or test_wave_split_kernel_vs_oracle" > $O/pytest_ws.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ws.txt
tail -5 $O/pytest_ws.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
