#!/bin/bash
# GPU box, round 4: typed frames (4 bytes in, 8 bytes out per sample: two thirds of the traffic are stores) -- store policies, and what the chip gives a
# kernel that only writes / reads 1 : writes 2 (torch fill_ and a strided copy as yardsticks).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04p; mkdir -p $O
MODE=${1:-run}
LG=8912896; LGP=8912928; NT=$((7<<16)); PL=$((1<<16)); SC=$((6<<16))
if [ "$MODE" = prebuild ]; then S="python tools/sweep.py --prebuild"; else S="timeout 600 python tools/sweep.py --rounds 7"; fi
export FLOWZ_HIP_AUTOTUNE=0 FLOWZ_HIP_NO_PLAN_CACHE=1
sweeps() {
$S --graph c32onepole 0,0,0,0 4,1,1024,$((LGP+NT)) 4,1,1024,$((LGP+SC)) 2,2,1024,$LG 2,2,1024,$((LG+NT)) 2,1,1024,$LGP 2,1,1024,$((LGP+NT)) 4,2,512,$LG 4,2,512,$((LG+NT))
$S --graph f64biquad 0,0,0,0 4,1,1024,$((LGP+NT)) 2,2,1024,$LG 2,2,1024,$((LG+NT)) 2,1,1024,$((LGP+NT))
}
if [ "$MODE" = prebuild ]; then sweeps; exit 0; fi
python - > $O/yardsticks.txt 2>&1 <<'PY'
import torch
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 1 << 32                                     # 16 GiB of float32
a = torch.empty(N, dtype=torch.float32, device="cuda"); b = torch.empty(2 * N, dtype=torch.float32, device="cuda")
a.normal_()
ms = timed(lambda: b.fill_(1.0)); print(f"write only (fill_ 32 GiB): {ms:.3f} ms  {8 * N / ms / 1e6:.0f} GB/s")
ms = timed(lambda: a.fill_(1.0)); print(f"write only (fill_ 16 GiB): {ms:.3f} ms  {4 * N / ms / 1e6:.0f} GB/s")
ms = timed(lambda: torch.sum(a)); print(f"read only (sum 16 GiB): {ms:.3f} ms  {4 * N / ms / 1e6:.0f} GB/s")
ms = timed(lambda: b[:N].copy_(a)); print(f"copy 16 GiB: {ms:.3f} ms  {8 * N / ms / 1e6:.0f} GB/s")
bv = b.view(N, 2)
ms = timed(lambda: torch.stack((a, a), dim=1, out=bv)); print(f"read 1 : write 2 (stack, 16 GiB -> 32 GiB): {ms:.3f} ms  {12 * N / ms / 1e6:.0f} GB/s")
PY
cat $O/yardsticks.txt
sweeps > $O/sweeps.txt 2>&1
grep -v amdgpu.ids $O/sweeps.txt
