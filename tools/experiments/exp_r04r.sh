#!/bin/bash
# GPU box, round 4: (1) fuzz run on the tree with the hold body / store-merge flag / wide lockstep frames; (2) store policies of the stream-major bodies;
# (3) what a plain copy of the few-stream blocks' bytes takes (the yardstick next to cascade6_16384 / _32768 / config 2).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04r; mkdir -p $O
MODE=${1:-run}
NT=$((7<<16)); PL=$((1<<16)); SC1=$((3<<16))
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
FLOWZ_HIP_AUTOTUNE=0 $S --sm 0,0,0,0 2,64,256,$((256+NT)) 2,64,256,$((256+PL)) 2,64,256,$((256+SC1))
FLOWZ_HIP_AUTOTUNE=0 $S --sm --graph par4 0,0,0,0 1,32,256,$NT 1,32,256,$PL 1,32,256,$SC1
FLOWZ_HIP_AUTOTUNE=0 $S --sm --graph osc 0,0,0,0 2,64,256,$((256+NT)) 1,128,256,264
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
timeout 420 python tools/fuzz_gpu.py 910000 4000 330 > $O/fuzz_gpu.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_gpu.txt
tail -4 $O/fuzz_gpu.txt
python - > $O/small_copies.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from zignal_amd import flowz as F
print("# fz_copy_kernel (float4, nt loads, nt|sc1 stores) on the bytes of a 6-biquad block x 4096 samples: src -> dst of n_streams x 4096 floats, back to back launches")
for ns in (8192, 16384, 32768, 65536, 131072, 262144, 1 << 20):
    n = ns * 4096
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_(); b = torch.empty_like(a)
    F.copy_probe(a, b); torch.cuda.synchronize()
    reps = max(5, int(2e9 / (8 * n)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): F.copy_probe(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{ns:8d} streams: {8 * n / 1e9:7.3f} GB  {ms:.4f} ms per launch  {8 * n / ms / 1e6:7.1f} GB/s  {8 * n / ms / 1e6 / 8000:.4f} of 8 TB/s")
PY
cat $O/small_copies.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
