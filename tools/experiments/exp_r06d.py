#!/usr/bin/env python3
"""Round 6 (GPU box): the LDS-ring combs in lockstep (the library default at 1 M streams x 4096) on six fresh allocations, under whatever kernel
experiment switches FLOWZ_HIP_EXTRA_OPTS holds (store / load cache policies), plus neighbouring chunk lengths and a third chunk buffer.
   usage: FLOWZ_HIP_EXTRA_OPTS="-DFZ_DBG_AUX_ST=2" exp_r06d.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F, workloads as W  # noqa: E402

L, G, P3 = F.C.FZ_VF_LOCKSTEP, F.C.FZ_VF_GRID_SYNC, F.C.FZ_VF_PREFETCH3
ns, T = 1 << 20, 4096
prog = F.compile(F.from_sexpr(W.lds_ring_comb()))
V = {"default": None, "u12": (1, 12, 256, L | G), "u20": (1, 20, 256, L | G), "free u32": (1, 32, 256, 0)}
if not os.environ.get("FLOWZ_HIP_EXTRA_OPTS"):
    V["u16 three buffers"] = (1, 16, 256, L | G | P3)
if os.environ.get("R06D_THREE_BUFFERS"):                     # second pass: is it the third buffer? other chunk lengths with three buffers, two buffers next to them
    V = {"default": None, "u16 two buffers": (1, 16, 256, L | G), "u16 three buffers": (1, 16, 256, L | G | P3), "u12 three buffers": (1, 12, 256, L | G | P3),
         "u20 three buffers": (1, 20, 256, L | G | P3), "u24 three buffers": (1, 24, 256, L | G | P3), "u8 three buffers": (1, 8, 256, L | G | P3)}
keep, rows = [], {k: [] for k in V}
rows = {k: [] for k in V}
b_alg = ns * (8 * T + 8 * prog.n_state)
for trial in range(6):
    if trial:
        keep.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    F.synth_fill(x, 20160512)
    st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
    runs = {}
    for k, v in V.items():
        vv = F.make_variant(*v) if v else None
        try:
            prog.run_block(x, state=st, out=y, variant=vv)
            runs[k] = (lambda vv=vv: prog.run_block(x, state=st, out=y, variant=vv))
        except F.FlowzError as e:
            rows[k].append("refused: " + str(e)[:50])
    torch.cuda.synchronize()
    tw = time.time()
    while time.time() - tw < 0.3:
        runs["default"]()
        torch.cuda.synchronize()
    times = {k: [] for k in runs}
    for _ in range(3):
        for k, fn in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5)
    for k in runs:
        rows[k].append(round(b_alg / sorted(times[k])[1] / 1e6 / 8000, 4))
    del x, y, st, runs
    torch.cuda.empty_cache()
print(json.dumps({"extra_opts": os.environ.get("FLOWZ_HIP_EXTRA_OPTS", ""), "kernel": prog.kernel_name(None, ns, T), **rows}), flush=True)
