#!/bin/bash
# GPU box, round 4: (1) rows off the 16-byte grid (odd stream counts): the lane's streams 64 apart (dword accesses, FZ_VF_RAGGED as
# of this round) against adjacent streams (b128 accesses, -DFZ_DBG_NO_STRIDED); (2) the oscillator chain on plain time-major frames
# with four streams per lane in 512-lane workgroups (256 registers per lane, two laps) against the default (two per lane, 1024 lanes).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04h2; mkdir -p $O
MODE=${1:-run}
L=524288; LG=8912896; LGP=8912928; R=268435456
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
odd() {
$S --streams 1048577 0,0,0,0 4,1,1024,$LGP 2,2,1024,$LG
$S --streams 1000001 0,0,0,0 4,1,1024,$LGP 2,2,1024,$LG
$S --streams 1000002 0,0,0,0 4,1,1024,$LGP 2,2,1024,$LG
$S --graph df1 --streams 1048577 0,0,0,0 4,1,1024,$LGP
}
osc() {
$S --graph osc 0,0,0,0 2,1,1024,$LGP 4,1,512,$LGP 4,1,512,$LG 4,2,512,$LG 2,2,512,$LG 1,4,1024,$((LG+8))
$S --graph par4 0,0,0,0 1,4,1024,$LG 1,2,1024,$LG 2,2,512,$LG 2,1,512,$LGP
}
# (3) typed frames (8 bytes per stream out): four streams per lane store a lane's 32 bytes as two b128 pieces, each half of every 32-byte
# sector -- with the write-through policy of the frame stores (nt | sc1) PMC shows twice the algorithmic write traffic; store policies nt
# alone (7 << 16) and plain (1 << 16), and two streams per lane (one b128 per lane)
typed() {
NT=$((7<<16)); PL=$((1<<16))
for g in c32onepole f64biquad; do
$S --graph $g 0,0,0,0 4,1,1024,$LGP 4,1,1024,$((LGP+NT)) 4,1,1024,$((LGP+PL)) 2,2,1024,$LG 2,2,1024,$((LG+NT)) 2,16,256,0 2,16,256,$NT
done
}
if [ "$MODE" = prebuild ]; then odd; FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_NO_STRIDED odd; osc; typed; exit 0; fi
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "ragged or laps_remainders or lockstep" > $O/pytest_ragged.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ragged.txt
tail -5 $O/pytest_ragged.txt
{ echo "## strided (the lane's streams 64 apart)"; odd; echo "## -DFZ_DBG_NO_STRIDED (adjacent streams, b128 / b64 accesses)"; FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_NO_STRIDED odd; echo "## osc / par4"; osc; echo "## typed frames"; typed; } > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
