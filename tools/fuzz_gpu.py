#!/usr/bin/env python3
"""Dev tool (GPU box): a long randomized parity run -- random (typed) graphs, kernels vs the oracle.
usage: tools/fuzz_gpu.py <first_seed> <count> [time limit in seconds: stops early, still prints the summary]"""
import time
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
    a, b = np.ascontiguousarray(a, dt), np.ascontiguousarray(b, dt)
    nan = np.isnan(a) & np.isnan(b)
    u = np.uint32 if np.dtype(dt) == np.float32 else np.uint64
    return np.array_equal(np.where(nan, 0, a).view(u), np.where(nan, 0, b).view(u))


TRACE = bool(os.environ.get("FUZZ_TRACE"))      # print every launch before it runs and synchronise after it: pins a faulting kernel


def trace(*what):
    if TRACE:
        torch.cuda.synchronize()
        print("   ", *what, flush=True)


first, count = int(sys.argv[1]), int(sys.argv[2])
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t_start, last = time.time(), first - 1
ns, T = int(os.environ.get("FUZZ_NS", "200")), 61      # (FUZZ_NS=512: whole waves, so that the lane groups take part)
ok = bad = skipped = refused = 0
for seed in range(first, first + count):
    if time.time() - t_start > limit:
        break
    last = seed
    if (seed - first) % 50 == 0:
        print(f"... seed {seed}: {ok} identical, {bad} mismatching, {skipped} skipped so far ({time.time() - t_start:.0f} s)", flush=True)
    for typed in (False, True):
        g, n_in = (R.make_typed(seed)[:2] if typed else (R.make_cmp(seed)[:2] if os.environ.get("FUZZ_CMP") else R.make(seed)[:2]))   # FUZZ_CMP=1: comparison / logical operators among the arithmetic (round 6)
        trace("seed", seed, "typed" if typed else "plain", g)
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
            LG = F.C.FZ_VF_LOCKSTEP | F.C.FZ_VF_GRID_SYNC            # round 3: lockstep workgroups, XCD-wide synchronised (ragged 200-stream grids)
            fl = int(rng.choice([0, F.C.FZ_VF_PREFETCH3, F.C.FZ_VF_MAX_WG(2), F.C.FZ_VF_NO_STAGE_PACK, F.C.FZ_VF_LOCKSTEP, LG, LG | F.C.FZ_VF_PREFETCH3]))
            blk = int(rng.choice([64, 128])) if (fl & F.C.FZ_VF_LOCKSTEP) else 256   # (small workgroups: several of them for 200 streams, so that barriers and counters do something)
            trace("P", P, "U", U, "block", blk, "flags", fl)
            try:
                y, _ = p.run_block(xd, variant=F.make_variant(P, U, blk, fl))
            except F.FlowzError as e:
                if (fl & F.C.FZ_VF_LOCKSTEP) and e.code == F.C.FZ_E_UNSUPPORTED:     # an explicit lockstep variant whose kernel has scratch is refused
                    refused += 1
                    continue
                raise
            res.append((f"P={P} U={U} block={blk} flags={fl}", same(y.cpu().numpy(), want, np.float32)))
        trace("f64 frames")
        y64, _ = p.run_block(xd, out_f64=True)
        res.append(("f64 frames", same(y64.cpu().numpy(), want64, np.float64)))
        cut = int(rng.integers(1, T))
        trace("chained at", cut)
        ya, st = p.run_block(xd[:cut].contiguous())
        yb, _ = p.run_block(xd[cut:].contiguous(), state=st)
        res.append((f"chained at {cut}", same(torch.cat([ya, yb]).cpu().numpy(), want, np.float32)))
        # the stream-major kernel on [stream][t][wire] buffers (first 60 samples: rows % 4 == 0)
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x[:60], (1, 0, 2)))).cuda()
        for P in (1, 2):
            U = int(rng.choice([4, 8, 16, 32]))
            trace("stream-major P", P, "U", U)
            try:
                ys, _ = p.run_block_stream_major(xs, variant=F.make_variant(P, U))
            except F.FlowzError as e:                     # two streams per lane with patches too large for one wave per SIMD: refused
                if P == 2 and e.code == F.C.FZ_E_UNSUPPORTED:
                    continue
                raise
            res.append((f"stream-major P={P} U={U}", same(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want[:60], np.float32)))
        # the long-run body of the stream-major kernel (1-in/1-out graphs, >= 256 samples), automatic and 64-sample phases
        if p.n_in == 1 and p.n_out == 1 and p.n_lds_slots == 0:
            TL = 264 + 4 * int(rng.integers(0, 20))
            xl = O.synth_input(seed + 1, np.arange(ns), TL, n_wires=1)
            wl = O.compile(g, ns).run(xl)
            xsl = torch.from_numpy(np.ascontiguousarray(np.transpose(xl, (1, 0, 2)))).cuda()
            for v in (None, F.make_variant(1, 64, 0, F.C.FZ_VF_SM_LONG), F.make_variant(2, 64, 0, F.C.FZ_VF_SM_LONG)):   # (the last: the pair body, two streams per lane)
                label = "auto" if v is None else f"P={v.streams_per_lane} U=64"
                trace("stream-major long", label, "T", TL)
                ys, _ = p.run_block_stream_major(xsl, variant=v)
                res.append((f"stream-major long {label} T={TL}", same(ys.permute(1, 0, 2).contiguous().cpu().numpy(), wl, np.float32)))
        # fz_compile_typed: ResultType through inputs (random float / double wires), state and outputs
        dts = [str(rng.choice(["f32", "f64"])) for _ in range(n_in)]
        try:
            ot = O.compile(g, ns, typed=True, in_dtypes=dts)
        except O.GraphError:
            ot = None
        try:
            pt = F.compile(F.from_sexpr(g), in_dtypes=dts)
        except F.FlowzError:
            pt = None
        if (ot is None) != (pt is None):
            res.append((f"typed compile accepted by one side only (oracle {ot is not None}, product {pt is not None}) {dts}", False))
        elif pt is not None:
            wires = [x[:, :, i].astype(np.float64 if dts[i] == "f64" else np.float32) * (1.0 + (1e-9 if dts[i] == "f64" else 0.0)) for i in range(n_in)]
            wantt = O.run_typed(ot, wires, T=T)
            trace("typed", dts)
            yt, _ = pt.run_block(torch.from_numpy(F.pack_typed(wires, dts)).cuda(), variant=F.make_variant(int(rng.choice([1, 2, 4])), 8))
            gott = F.unpack_typed(yt.cpu().numpy(), pt.output_dtypes())
            okt = len(gott) == len(wantt)
            for a, b in zip(gott, wantt):
                if a.dtype != b.dtype:
                    okt = False
                elif a.dtype in (np.complex64, np.complex128):
                    part = np.float32 if a.dtype == np.complex64 else np.float64
                    okt = okt and same(a.real, b.real, part) and same(a.imag, b.imag, part)
                else:
                    okt = okt and same(a, b, a.dtype.type)
            res.append((f"typed {dts}", okt))
        trace("done")
        if all(r for _, r in res):
            ok += 1
        else:
            bad += 1
            print("MISMATCH seed", seed, "typed" if typed else "plain", [n for n, r in res if not r], g, flush=True)
print(f"fuzz seeds {first}..{last}: {ok} graphs identical, {bad} mismatching, {skipped} skipped, {refused} lockstep variants refused "
      f"(per graph: P = 1, 2, 4 with random unroll / flags incl. lockstep and XCD-synchronised workgroups, float64 frames, a split block, the stream-major kernel short, long and pair-long, fz_compile_typed with random input types)")
sys.exit(1 if bad else 0)
