#!/usr/bin/env python3
"""Experiment (GPU box): from which stream count on does the pair long-run stream-major body beat the one-stream body? (6-biquad cascade x 4096)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["FLOWZ_HIP_AUTOTUNE"] = "0"
import torch
from zignal_amd import workloads as G, flowz as F
def timed(fn, reps):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
for ns in (1 << 17, 1 << 18, 3 << 17, 1 << 19):
    T = 4096
    x = torch.randn((ns, T, 1), device="cuda") * 0.1
    out = torch.empty_like(x); st = torch.zeros((prog.n_state, ns), device="cuda")
    b = ns * T * 8
    for rnd in range(3):
        r = []
        for label, v in (("one", F.make_variant(1, 128, 0, 256)), ("pair", F.make_variant(2, 64, 0, 256))):
            ms = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v), 20)
            r.append(f"{label} {ms:.3f} ms ({b / ms / 8e9:.4f})")
        print(ns, " | ".join(r), flush=True)
