#!/usr/bin/env python3
"""Experiment (GPU box): how much of the stream-major kernel's time is arithmetic?  The same skeleton on graphs from a gain to six biquads."""
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


graphs = {"gain": lambda: G.mul(G.IN(1), G.lit(0.5)), "df1": G.df1, "cascade2": lambda: G.df1_cascade(2), "cascade3": lambda: G.df1_cascade(3),
          "cascade4": lambda: G.df1_cascade(4), "cascade6": lambda: G.df1_cascade(6)}
ns = 1 << 20
for T in (4096, 1024):
    x = torch.randn((ns, T, 1), device="cuda") * 0.1
    out = torch.empty((ns, T, 1), device="cuda")
    b = ns * T * 8
    for name, mk in graphs.items():
        prog = F.compile(F.from_sexpr(mk()))
        st = torch.zeros((prog.n_state, ns), device="cuda") if prog.n_state else None
        line = f"{name:9s} T={T}:"
        for P, U, fl in ((1, 128, 256), (1, 64, 256), (1, 128, 256 | 16), (0, 0, 512)):
            v = F.make_variant(P, U, 0, fl)
            try:
                ms = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
                line += f"  U={U} fl={fl}: {ms:6.3f} ms {b / ms / 1e6:6.0f} GB/s [{prog.kernel_name(v, ns, T, 0)[-24:]}]"
            except F.FlowzError as e:
                line += f"  U={U} fl={fl}: refused"
        print(line, flush=True)
    # the copy yardstick on the same bytes
    y = torch.empty_like(x)
    ms = timed(lambda: F.copy_probe(x, y))
    print(f"copy kernel T={T}: {ms:6.3f} ms {b / ms / 1e6:6.0f} GB/s", flush=True)
