#!/usr/bin/env python3
"""Dev tool (GPU box): time-major frames at 1 M streams -- one launch over 4096 samples against a loop of launches over
windows ("slabs") of 128 ... 1024 samples.  Rows of a time-major frame are 4 MB apart: the fewer rows the resident waves
are spread over, the fewer pages are in flight (translation misses)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F  # noqa: E402
from zignal_amd import workloads as W  # noqa: E402

ns, T = 1 << 20, 4096
prog = F.compile(F.from_sexpr(W.df1_cascade(6)))
x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
y = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
F.synth_fill(x, 1)
st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
b_alg = ns * (8 * T + 8 * prog.n_state)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for v in (None, F.make_variant(2, 16), F.make_variant(4, 8), F.make_variant(4, 4, 256, F.C.FZ_VF_MAX_WG(2))):
    name = "default" if v is None else f"{v.streams_per_lane},{v.unroll},{v.block_threads},{v.flags}"
    ms = timed(lambda: prog.run_block(x, state=st, out=y, variant=v))
    print(f"{name:24s} one launch            {ms:7.3f} ms  {b_alg / ms / 1e6:7.1f} GB/s  {b_alg / ms / 8e9:.3f}", flush=True)
    for slab in (128, 256, 512, 1024):
        def run():
            for r0 in range(0, T, slab):
                prog.run_window(x, y, st, r0, slab, variant=v)
        ms = timed(run)
        print(f"{name:24s} slabs of {slab:5d} samples {ms:7.3f} ms  {b_alg / ms / 1e6:7.1f} GB/s  {b_alg / ms / 8e9:.3f}", flush=True)
