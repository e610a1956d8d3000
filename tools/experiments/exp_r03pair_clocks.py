#!/usr/bin/env python3
"""Experiment (GPU box): FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_PHASE_CLOCKS -- where one wave of the PAIR long-run stream-major kernel spends its clocks."""
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
names = ["wait for an in-run (two per phase)", "park it (64 ds_write2_b32), twice", "request the next (32 buffer loads), twice", "2 x 64 steps of two streams",
         "keep the first half's outputs (64 ds_read2_b32)", "out-run (64 ds_read2_b32 + 64 stores)"]
for name, mk in (("cascade6", lambda: G.df1_cascade(6)), ("cascade2", lambda: G.df1_cascade(2))):
    prog = F.compile(F.from_sexpr(mk()))
    st = torch.zeros((prog.n_state, ns), device="cuda")
    v = F.make_variant(2, 64, 0, 256)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prog.run_block_stream_major(x, state=st, out=out, variant=v)
        e1.record()
        torch.cuda.synchronize()
    c = out.view(torch.int32)[0, :7, 0].cpu().numpy().astype("uint32")
    tot = int(c[1:].sum())
    nph = T // 128
    print(f"{name} pair body: kernel {e0.elapsed_time(e1):.3f} ms; one wave, {nph} phases of 128 samples x 128 streams, {tot} clocks of s_memtime (100 MHz) -> per phase:")
    for k in range(6):
        print(f"    {names[k]:52s} {c[k + 1] / nph:10.1f}  ({100.0 * c[k + 1] / max(tot, 1):5.1f} %)")
