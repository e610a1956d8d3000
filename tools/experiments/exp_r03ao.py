#!/usr/bin/env python3
"""Experiment (GPU box): 4096-sample blocks of few streams back to back -- eager launches against one captured hipGraph of 20 of them
(HIP events around both)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zignal_amd import workloads as G, flowz as F
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
T, K = 4096, 20
for ns in (16384, 32768, 65536, 1 << 20):
    tile = 8192
    x = torch.randn((ns // tile, T, tile, 1), device="cuda") * 0.1
    y = torch.empty_like(x)
    st = torch.zeros((prog.n_state, ns), device="cuda")
    def blocks():
        for _ in range(K): prog.run_block(x, state=st, out=y)
    blocks(); torch.cuda.synchronize()
    def ev(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best / K
    t_e = ev(blocks)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            blocks()
    t_g = ev(g.replay)
    b = ns * (8 * T + 8 * 14)
    print(f"{ns:8d} streams: eager {t_e * 1e3:7.1f} us/launch ({b / t_e / 1e6 / 8000:.4f} of peak), hipGraph {t_g * 1e3:7.1f} us/launch ({b / t_g / 1e6 / 8000:.4f})", flush=True)
