#!/usr/bin/env python3
"""Dev tool (GPU box): randomized parity run for WIDE frames (three to eight input wires) -- the bodies round 4 added: the stream-major short-chunk
body that holds its outputs (n_in a multiple of 4 and of n_out), and the frame kernel of wide frames in lockstep / XCD-synchronised workgroups.
Random graphs from tests/randgraphs.py with 3..8 input wires, random stream counts (ragged), block lengths and chunk depths; vs the oracle.
usage: tools/fuzz_wide_frames.py <first_seed> <count> [time limit in seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import randgraphs as R  # noqa: E402
from oracle import flowz_oracle as O  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(nan, 0, a).view(np.uint32), np.where(nan, 0, b).view(np.uint32))


first, count = int(sys.argv[1]), int(sys.argv[2])
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t0, ok, bad, skipped, held, last = time.time(), 0, 0, 0, 0, first - 1
LG = F.C.FZ_VF_LOCKSTEP | F.C.FZ_VF_GRID_SYNC
for seed in range(first, first + count):
    if time.time() - t0 > limit:
        break
    last = seed
    rng = np.random.default_rng(seed)
    n_in = int(rng.choice([3, 4, 4, 4, 5, 8, 8]))
    g, n_out = R.graph(rng, n_in, int(rng.integers(1, 4)))
    ns = int(rng.choice([1, 63, 64, 130, 200, 777, 1000, 2050]))
    T = 4 * int(rng.integers(1, 90))
    try:
        f = O.compile(g, ns)
    except O.GraphError:
        skipped += 1
        continue
    x = O.synth_input(seed, np.arange(ns), T, n_wires=n_in)
    want = f.run(x)
    p = F.compile(F.from_sexpr(g))
    hold = n_in % 4 == 0 and n_out < n_in and n_in % n_out == 0
    held += hold
    res = []
    xd = torch.from_numpy(x).cuda()
    y0, st0 = p.run_block(xd, variant=F.make_variant(1, 8, 256, F.C.FZ_VF_NO_STAGE_PACK))
    res.append(("frames P=1", same(y0.cpu().numpy(), want)))
    for v in ((1, int(rng.choice([1, 2, 4])), int(rng.choice([64, 256, 1024])), LG), (1, 1, 128, LG | F.C.FZ_VF_PREFETCH3)):
        try:
            y, st = p.run_block(xd, variant=F.make_variant(*v))
        except F.FlowzError as e:
            if e.code == F.C.FZ_E_UNSUPPORTED:        # (a kernel with scratch in a lockstep workgroup is refused)
                continue
            raise
        res.append((f"lockstep {v}", torch.equal(y, y0) and torch.equal(st, st0)))
    xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
    for U in (0, int(rng.choice([4, 8, 16, 32]))):
        try:
            ys, sts = p.run_block_stream_major(xs, variant=F.make_variant(1, U) if U else None)
        except F.FlowzError as e:
            if e.code == F.C.FZ_E_UNSUPPORTED:        # (patches that do not fit)
                continue
            raise
        res.append((f"stream-major U={U}{' (held outputs)' if hold else ''}", same(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want) and torch.equal(sts, st0)))
    # two windows of the stream-major buffers, state carried (an out-run that starts off the run grid)
    if T >= 16:
        cut = 4 * int(rng.integers(1, T // 4))
        out = torch.zeros((ns, T, n_out), device="cuda")
        _, s1 = p.run_block_stream_major(xs, out=out, n_samples=cut)
        p.run_block_stream_major(xs, out=out, state=s1, row0=cut)
        res.append((f"stream-major windows at {cut}", same(out.permute(1, 0, 2).contiguous().cpu().numpy(), want)))
    if all(r for _, r in res):
        ok += 1
    else:
        bad += 1
        print("MISMATCH seed", seed, f"n_in={n_in} n_out={n_out} ns={ns} T={T}", [n for n, r in res if not r], g, flush=True)
    if (seed - first) % 25 == 24:
        print(f"... seed {seed}: {ok} identical, {bad} mismatching ({time.time() - t0:.0f} s)", flush=True)
print(f"wide-frame fuzz seeds {first}..{last}: {ok} graphs identical, {bad} mismatching, {skipped} skipped; {held} of them with held outputs "
      f"(per graph: frame kernel, two lockstep / XCD-synchronised variants, stream-major default + a random chunk depth, two stream-major windows)")
sys.exit(1 if bad else 0)
