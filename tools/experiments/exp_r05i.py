#!/usr/bin/env python3
"""Round 5 (GPU box): do odd chunk lengths make the FREE-RUNNING frame kernels less dependent on where the buffers lie?  The 6-biquad cascade (two streams per
lane, 256-lane workgroups, no lockstep) and the LDS-ring combs (one stream per lane) at 1 M streams x 4096 on plain time-major rows, eight fresh allocations
each (other allocations of odd sizes in between), chunks of 12 ... 32 rows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F, workloads as W  # noqa: E402

ns, T = 1 << 20, 4096
cases = {"cascade6": (W.df1_cascade(6), [(2, u, 256) for u in (16, 15, 17, 12, 20, 24)]),
         "ldsring": (W.lds_ring_comb(), [(1, 32, 256), (1, 24, 256), (1, 16, 256), (1, 48, 256)])}
which = sys.argv[1:] or list(cases)
prebuild = bool(os.environ.get("PREBUILD"))
for name in which:
    g, variants = cases[name]
    prog = F.compile(F.from_sexpr(g))
    if prebuild:
        for v in variants:
            try:
                prog.build(F.make_variant(*v), ns, T)
            except F.FlowzError as e:
                print("refused", v, str(e)[:80])
        continue
    keep = []
    b = ns * (8 * T + 8 * prog.n_state)
    for trial in range(8):
        if trial:
            keep.append(torch.empty(((trial * 29 + 7) << 20,), dtype=torch.uint8, device="cuda"))
        x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
        y = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
        F.synth_fill(x, 20160512)
        st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
        row = {"graph": name, "trial": trial, "x-y mod 16MiB (MiB)": ((x.data_ptr() - y.data_ptr()) % (16 << 20)) / (1 << 20)}
        for v in variants:
            try:
                vv = F.make_variant(*v)
                prog.run_block(x, state=st, out=y, variant=vv)
            except F.FlowzError:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                prog.run_block(x, state=st, out=y, variant=vv)
            e1.record()
            torch.cuda.synchronize()
            row[f"u{v[1]}"] = round(b / (e0.elapsed_time(e1) / 5) / 1e6 / 8000, 3)
        print(json.dumps(row), flush=True)
        del x, y, st
        torch.cuda.empty_cache()
