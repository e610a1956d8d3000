#!/usr/bin/env python3
"""Dev tool (GPU box): a long randomized parity run -- random (typed) graphs, kernels vs the oracle.
usage: tools/fuzz_gpu.py <first_seed> <count>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import randgraphs as R  # noqa: E402
from oracle import flowz_oracle as O  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402


def same(a, b, dt):
    a, b = np.asarray(a, dt), np.asarray(b, dt)
    nan = np.isnan(a) & np.isnan(b)
    u = np.uint32 if dt == np.float32 else np.uint64
    return np.array_equal(np.where(nan, 0, a).view(u), np.where(nan, 0, b).view(u))


first, count = int(sys.argv[1]), int(sys.argv[2])
ns, T = 200, 61
ok = bad = skipped = 0
for seed in range(first, first + count):
    for typed in (False, True):
        g, n_in = (R.make_typed(seed)[:2] if typed else R.make(seed)[:2])
        try:
            f = O.compile(g, ns)
            f64 = O.compile(g, ns, out_f64=True)
        except O.GraphError:
            skipped += 1
            continue
        x = O.synth_input(seed, np.arange(ns), T, n_wires=n_in)
        want, want64 = f.run(x), f64.run(x)
        p = F.compile(F.from_sexpr(g))
        xd = torch.from_numpy(x).cuda()
        res = []
        rng = np.random.default_rng(seed)
        for P in (1, 2, 4):
            U = int(rng.choice([1, 3, 8, 16]))
            fl = int(rng.choice([0, F.C.FZ_VF_PREFETCH3, F.C.FZ_VF_MAX_WG(2), F.C.FZ_VF_NO_STAGE_PACK]))
            y, _ = p.run_block(xd, variant=F.make_variant(P, U, 256, fl))
            res.append((f"P={P} U={U} flags={fl}", same(y.cpu().numpy(), want, np.float32)))
        y64, _ = p.run_block(xd, out_f64=True)
        res.append(("f64 frames", same(y64.cpu().numpy(), want64, np.float64)))
        cut = int(rng.integers(1, T))
        ya, st = p.run_block(xd[:cut].contiguous())
        yb, _ = p.run_block(xd[cut:].contiguous(), state=st)
        res.append((f"chained at {cut}", same(torch.cat([ya, yb]).cpu().numpy(), want, np.float32)))
        # the stream-major kernel on [stream][t][wire] buffers (first 60 samples: rows % 4 == 0)
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x[:60], (1, 0, 2)))).cuda()
        for P in (1, 2):
            U = int(rng.choice([4, 8, 16, 32]))
            ys, _ = p.run_block_stream_major(xs, variant=F.make_variant(P, U))
            res.append((f"stream-major P={P} U={U}", same(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want[:60], np.float32)))
        if all(r for _, r in res):
            ok += 1
        else:
            bad += 1
            print("MISMATCH seed", seed, "typed" if typed else "plain", [n for n, r in res if not r], g, flush=True)
print(f"fuzz seeds {first}..{first + count - 1}: {ok} graphs identical, {bad} mismatching, {skipped} skipped")
sys.exit(1 if bad else 0)
