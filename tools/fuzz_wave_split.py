#!/usr/bin/env python3
"""Dev tool (GPU box): randomized parity run of the wave-split / I/O-wave kernels -- random serial filters, random sizes, every split
the graph allows with random unroll / workgroup size / I/O wave, two chained blocks -- against the oracle.
usage: tools/fuzz_wave_split.py <first_seed> <count> [time limit in seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import graphs as G  # noqa: E402
from oracle import flowz_oracle as O  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t0, ok, bad, launches = time.time(), 0, 0, 0
for seed in range(first, first + count):
    if time.time() - t0 > limit:
        break
    rng = np.random.default_rng(20000 + seed)
    form = ["df1", "df2", "df1t"][seed % 3]
    n = int(rng.choice([4, 6, 8, 10, 12, 16]))

    def stage():
        if form == "df1t":
            return G.df1t()
        r, th = rng.uniform(0.3, 0.95), rng.uniform(0.1, 3.0)
        c = (rng.uniform(0.1, 1.0), rng.uniform(-1, 1), rng.uniform(-1, 1), 2 * r * np.cos(th), -r * r)
        return (G.df1 if form == "df1" else G.df2)(*[float(np.float32(v)) for v in c])

    g = stage()
    for _ in range(n - 1):
        g = G.seq(g, stage())
    r = rng.random()
    if r < 0.2:                                            # a scalar prefix in front of the chain
        g = G.seq(G.fb(G.add(G.mul(G.lit(float(np.float32(rng.uniform(0.1, 0.6)))), G.DEL(1, 1)), G.IN(2))), g)
    elif r < 0.4:                                          # a scalar suffix behind it
        g = G.seq(g, G.mul(G.lit(float(np.float32(rng.uniform(0.2, 1.5)))), G.IN(1)))
    elif r < 0.5:                                          # both
        g = G.seq(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))), G.seq(g, G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.mul(G.lit(0.5), G.IN(2))))))
    prog = F.compile(F.from_sexpr(g))
    ns, T = int(rng.integers(1, 700)), int(rng.integers(1, 1500))
    x = O.synth_input(seed, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    xd = torch.from_numpy(x).cuda()
    good = True
    for W in (1, 2, 3, 4):
        io = F.C.FZ_VF_IO_WAVE if (W == 1 or rng.random() < 0.5) else 0
        v = F.make_variant(1, int(rng.choice([8, 16, 32])), int(rng.choice([0, 64, 128])), F.C.FZ_VF_WAVES(W) | io)
        try:
            prog.kernel_name(v, ns, T)
        except F.FlowzError:
            continue
        cut = int(rng.integers(0, T + 1))
        st, parts = None, []
        for lo, hi in ((0, cut), (cut, T)):
            if hi > lo:
                y, st = prog.run_block(xd[lo:hi].clone(), state=st, variant=v)       # (a fresh, aligned buffer)
                parts.append(y.cpu().numpy())
                launches += 1
        got = np.concatenate(parts)
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            good = False
            print("MISMATCH seed", seed, form, n, ns, T, "W", W, "io", bool(io), "cut", cut, flush=True)
    ok, bad = ok + good, bad + (not good)
    if (seed - first) % 50 == 0:
        print(f"... seed {seed}: {ok} identical, {bad} mismatching ({time.time() - t0:.0f} s)", flush=True)
print(f"wave-split fuzz seeds {first}..{seed}: {ok} cascades identical, {bad} mismatching, {launches} launches (parts 1-4, with and without the I/O wave, unroll 8/16/32, two chained blocks)")
