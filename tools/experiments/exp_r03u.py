#!/usr/bin/env python3
"""Experiment (GPU box): the stream-major long-run kernel with its runs sent through zero-byte descriptors
(FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_NOLOAD / -DFZ_DBG_NOSTORE): what the lone wave (U=128) and the two waves per SIMD (U=64) cost without memory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402


def timed(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ns, T = 1 << 20, 4096
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
b = ns * T * 8
for name, mk in (("cascade6", lambda: G.df1_cascade(6)), ("cascade4", lambda: G.df1_cascade(4)), ("gain", lambda: G.mul(G.IN(1), G.lit(0.5)))):
    prog = F.compile(F.from_sexpr(mk()))
    st = torch.zeros((prog.n_state, ns), device="cuda") if prog.n_state else None
    line = f"{name:9s} opts='{os.environ.get('FLOWZ_HIP_EXTRA_OPTS', '')}':"
    for U in (128, 64):
        v = F.make_variant(1, U, 0, 256)
        ms = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
        line += f"  U={U}: {ms:6.3f} ms ({b / ms / 1e6:5.0f} GB/s equiv.)"
    print(line, flush=True)
