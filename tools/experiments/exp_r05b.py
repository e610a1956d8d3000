#!/usr/bin/env python3
"""Round 5 (GPU box): does the rate of a kernel depend on WHERE its buffers were allocated?  The same typed kernels (complex one-pole,
1 M streams x 4096: 4 bytes in, 8 bytes out per sample) on six fresh allocations each, with other allocations of odd sizes in between:
the free-running two-streams-per-lane kernel and the lockstep four-streams-per-lane one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F, workloads as W  # noqa: E402

ns, T = 1 << 20, 4096
prog = F.compile(F.from_sexpr(W.complex_one_pole()), typed=True)
variants = {"p2u16b256 free": F.make_variant(2, 16, 256), "p4u1b1024 lockstep": F.make_variant(4, 1, 1024, 524288 | 8388608 | 32)}
keep = []
for trial in range(6):
    if trial:
        keep.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))      # shifts what the next allocations get
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    y = torch.empty((T, ns, prog.n_out), dtype=torch.float32, device="cuda")
    F.synth_fill(x, 20160512)
    st = torch.zeros((max(prog.n_state, 1), ns), dtype=torch.float32, device="cuda")
    row = {"trial": trial, "x_ptr": hex(x.data_ptr()), "y_ptr": hex(y.data_ptr())}
    for name, v in variants.items():
        prog.run_block(x, state=st, out=y, variant=v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            prog.run_block(x, state=st, out=y, variant=v)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row[name] = round(ns * (4 * T * 3 + 16) / ms / 1e6 / 8000, 4)
    print(json.dumps(row), flush=True)
    del x, y, st
    torch.cuda.empty_cache()
