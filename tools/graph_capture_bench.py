#!/usr/bin/env python3
"""Dev tool (GPU box): short blocks back to back, eager launches vs one captured hipGraph of them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
for ns, T, nblk in ((65536, 64, 32), (65536, 256, 32), (8192, 64, 32), (1 << 20, 64, 16)):
    tile = 8192 if ns > 8192 else 0
    shape = (nblk, ns // tile, T, tile, 1) if tile else (nblk, T, ns, 1)
    x = torch.empty(shape, device="cuda")
    F.synth_fill(x.view(-1, *shape[-2:]) if not tile else x.view(-1, T, tile, 1), 1)
    y = torch.empty_like(x)
    st = torch.zeros((prog.n_state, ns), device="cuda")
    def blocks():
        for k in range(nblk):
            prog.run_block(x[k], state=st, out=y[k])
    blocks()                                        # JIT, module load
    torch.cuda.synchronize()
    st.zero_(); blocks(); torch.cuda.synchronize(); ref = y.clone(); st_ref = st.clone()
    def timed(fn, reps=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    t_eager = timed(blocks)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            blocks()
    st.zero_(); y.zero_(); g.replay(); torch.cuda.synchronize()
    same = torch.equal(y, ref) and torch.equal(st, st_ref)
    t_graph = timed(g.replay)
    print(f"{ns} streams x {T} samples x {nblk} blocks: eager {t_eager * 1e6 / nblk:7.1f} us/block, hipGraph {t_graph * 1e6 / nblk:7.1f} us/block, "
          f"identical results: {same}")
