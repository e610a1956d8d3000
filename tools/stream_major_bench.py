#!/usr/bin/env python3
"""Dev tool (GPU box): callers with [stream][t] buffers -- the stream-major block kernel vs
adapter + frame kernel + adapter."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


graphs = {"cascade6": lambda: G.df1_cascade(6), "cascade2": lambda: G.df1_cascade(2), "df1": G.df1, "par4": G.par4_sum}
for name, ns, T in (("cascade6", 1 << 20, 4096), ("cascade6", 1 << 20, 1024), ("cascade2", 1 << 20, 1024), ("df1", 1 << 20, 1024), ("par4", 1 << 18, 1024),
                    ("cascade6", 65536, 4096)):
    prog = F.compile(F.from_sexpr(graphs[name]()))
    w = max(prog.n_in, 1)
    x = torch.randn((ns, T, w), device="cuda") * 0.1
    out = torch.empty((ns, T, prog.n_out), device="cuda")
    st = torch.zeros((prog.n_state, ns), device="cuda")
    tile = prog.recommended_tile_streams()
    fr = F.frames_from_stream_major(x, tile)
    yf = torch.empty((ns // tile, T, tile, prog.n_out), device="cuda") if tile < ns else torch.empty((T, ns, prog.n_out), device="cuda")
    b = ns * T * 4 * (prog.n_in + prog.n_out)
    res = {}
    for P, U, fl in ((0, 0, 0), (1, 128, 256), (1, 64, 256), (1, 128, 256 | 16), (0, 0, 512), (1, 32, 16), (2, 32, 0)):
        v = F.make_variant(P, U, 0, fl) if (P or U or fl) else None
        label = f"stream-major kernel {'auto' if v is None else f'P={P} U={U} flags={fl}'}"
        try:
            res[label] = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
        except F.FlowzError:
            res[label] = float("nan")

    def via_adapter():
        F.frames_from_stream_major(x, tile, out=fr)
        prog.run_block(fr, state=st, out=yf)
        F.frames_to_stream_major(yf, out=out)
    res["adapter + frame kernel + adapter"] = timed(via_adapter)
    res["frame kernel alone (frames resident)"] = timed(lambda: prog.run_block(fr, state=st, out=yf))
    print(f"# {name}, {ns} streams x {T} samples, B_alg {b / 1e9:.2f} GB")
    for k, ms in res.items():
        print(f"  {k:40s} {ms:8.3f} ms  {ns * T / ms / 1e3:9.1f} Msamples/s  {b / ms / 1e6:7.1f} GB/s algorithmic")
