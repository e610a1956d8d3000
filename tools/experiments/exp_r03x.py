#!/usr/bin/env python3
"""Experiment (GPU box): FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_PHASE_CLOCKS -- where one wave of the long-run stream-major kernel spends its clocks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["FLOWZ_HIP_EXTRA_OPTS"] = (os.environ.get("FLOWZ_HIP_EXTRA_OPTS", "") + " -DFZ_DBG_PHASE_CLOCKS").strip()
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

ns, T = 1 << 20, 4096
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
names = ["wait for the in-run", "park it (32 ds_write_b128)", "request the next (32 buffer loads)", "128 steps", "out-run (32 ds_read_b128 + 32 stores)"]
for name, mk in (("cascade6", lambda: G.df1_cascade(6)), ("cascade4", lambda: G.df1_cascade(4)), ("cascade2", lambda: G.df1_cascade(2))):
    prog = F.compile(F.from_sexpr(mk()))
    st = torch.zeros((prog.n_state, ns), device="cuda")
    for U in (128, 64):
        v = F.make_variant(1, U, 0, 256)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prog.run_block_stream_major(x, state=st, out=out, variant=v)
            e1.record()
            torch.cuda.synchronize()
        c = out.view(torch.int32)[0, :6, 0].cpu().numpy().astype("uint32")
        tot = int(c[1:].sum())
        nph = T // U
        print(f"{name} U={U} opts='{os.environ['FLOWZ_HIP_EXTRA_OPTS']}': kernel {e0.elapsed_time(e1):.3f} ms; one wave, {nph} phases, {tot} clocks (s_memtime: 100 MHz?) -> per phase:")
        for k in range(5):
            print(f"    {names[k]:40s} {c[k + 1] / nph:10.1f}  ({100.0 * c[k + 1] / max(tot, 1):5.1f} %)")
