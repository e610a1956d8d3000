#!/usr/bin/env python3
"""Dev tool (GPU box): time kernel variants of a workload with HIP events, interleaved rounds.

usage: tools/sweep.py [--graph cascade6] [--streams N] [--samples T] [--rounds R] variants...
       variant = P,U[,block[,flags]]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="cascade6")
    ap.add_argument("--streams", type=int, default=1 << 20)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=1, help="back-to-back launches per timing (sub-millisecond kernels: >= 20)")
    ap.add_argument("--tile", type=int, default=0, help="streams per frame tile (0 = time-major)")
    ap.add_argument("--sm", action="store_true", help="stream-major buffers [stream][t][wire] (fz_run_block_stream_major; FZ_VF_STREAM_MAJOR is added to the flags)")
    ap.add_argument("--prebuild", action="store_true", help="no GPU: build the variants' kernels into the cache (run without torch: the installation's compiler)")
    ap.add_argument("variants", nargs="*", default=["1,8", "2,8", "4,4"])
    a = ap.parse_args()
    import numpy as np
    if not a.prebuild:
        import torch

    from zignal_amd import workloads as G
    W = G
    from zignal_amd import flowz as F

    graphs = {"cascade6": lambda: G.df1_cascade(6), "par4": G.par4_sum, "par4f": G.par4_sum_fanout,
              "osc": lambda: G.osc_chain(6), "df1": G.df1, "df2": G.df2, "df1t": G.df1t, "df2t": G.df2t,
              "gain": lambda: G.mul(G.lit(0.5), G.IN(1)), "cascade2": lambda: G.df1_cascade(2), "cascade4": lambda: G.df1_cascade(4),
              "cascade6g": lambda: G.seq(G.df1_cascade(6), G.mul(G.lit(0.7), G.IN(1))),
              "cascade8": lambda: G.df1_cascade(8), "cascade10": lambda: G.df1_cascade(10), "cascade12": lambda: G.df1_cascade(12), "cascade24": lambda: G.df1_cascade(24),
              "mod6": lambda: G.df1_cascade_modulated(6), "ldsring": G.lds_ring_comb, "farring": lambda: G.far_comb(300), "ident": lambda: G.IN(1),
              "params6": lambda: G.df1_cascade_params(6), "c32onepole": G.complex_one_pole, "f64biquad": G.df1_double,
              # yardsticks of the typed frames: a float wire in, a double / complex<float> wire out, one operation (4 bytes read, 8 written per sample)
              "widen64": lambda: G.mul(G.lit64(1.0), G.IN(1)), "widenc32": lambda: G.mul(G.litc(1.0, 0.0), G.IN(1))}
    prog = F.compile(F.from_sexpr(graphs[a.graph]()), typed=a.graph in ("c32onepole", "f64biquad", "widen64", "widenc32"))
    ns, T = a.streams, a.samples
    if a.prebuild:
        for s in a.variants:
            if s == "tune":
                vs = prog.tune_candidates(ns, T, a.tile)
            else:
                t = [int(v) for v in s.split(",")]
                t += [0] * (4 - len(t))
                if a.sm:
                    t[3] |= 128
                vs = [F.make_variant(*t)]
            for v in vs:
                try:
                    prog.build(v, ns, T, a.tile)
                except F.FlowzError as e:
                    print(f"# {a.graph} {ns} {s}: refused: {str(e)[:100]}")
        return
    if a.sm:
        x = torch.empty((ns, T, max(prog.n_in, 1)), dtype=torch.float32, device="cuda")
        y = torch.empty((ns, T, prog.n_out), dtype=torch.float32, device="cuda")
    elif a.tile:
        x = torch.empty((ns // a.tile, T, a.tile, max(prog.n_in, 1)), dtype=torch.float32, device="cuda")
        y = torch.empty((ns // a.tile, T, a.tile, prog.n_out), dtype=torch.float32, device="cuda")
    else:
        x = torch.empty((T, ns, max(prog.n_in, 1)), dtype=torch.float32, device="cuda")
        y = torch.empty((T, ns, prog.n_out), dtype=torch.float32, device="cuda")
    if a.sm:
        x.normal_(0.0, 0.1)
    else:
        F.synth_fill(x, 20160512)
    run = (lambda v: prog.run_block_stream_major(x, state=state, params=params, out=y, variant=v)) if a.sm else \
          (lambda v: prog.run_block(x, state=state, params=params, out=y, variant=v))
    params = None
    if prog.n_param:
        P = W.osc_chain_params(20160513, np.arange(ns))
        params = torch.from_numpy(np.ascontiguousarray(P[:prog.n_param] if a.graph == "osc" else P[1:1 + prog.n_param])).cuda()
    if prog.n_mod:
        prog.set_modulation((0.2 + 0.1 * torch.rand((prog.n_mod, T), device="cuda")).contiguous())
    state = torch.zeros((max(prog.n_state, 1), ns), dtype=torch.float32, device="cuda")
    b_alg = ns * (4 * T * (prog.n_in + prog.n_out) + 8 * prog.n_state + 4 * prog.n_param)
    vs = []
    for s in a.variants:
        if s == "tune":                          # the plan fz_program_tune selects on this box
            cv, _ = prog.tune(x, state=state, params=params, out=y)
            vs.append((f"tune->{cv.streams_per_lane},{cv.unroll},{cv.block_threads},{cv.flags}", cv))
            continue
        t = [int(v) for v in s.split(",")]
        t += [0] * (4 - len(t))
        if a.sm:
            t[3] |= 128
        vs.append((s, F.make_variant(*t) if any(t[:3]) or (t[3] & ~128) else None))
    times = {s: [] for s, _ in vs}
    ok = []
    for s, v in vs:                              # warm (JIT + first touch); a variant the graph / shape refuses is reported and skipped
        try:
            run(v)
            ok.append((s + " " + prog.kernel_name(v if v is not None else (F.make_variant(0, 0, 0, 128) if a.sm else prog.plan(ns, a.tile)), ns, T, a.tile).replace("fz_block_kernel_", ""), v))
        except F.FlowzError as e:
            print(f"# {s}: refused: {str(e)[:120]}")
    vs = ok
    times = {s: [] for s, _ in vs}
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for s, v in vs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(a.reps):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / a.reps)
    # copy yardstick
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = min(x.numel(), y.numel())
    xs, ys = x.view(-1)[:n], y.view(-1)[:n]
    F.copy_probe(xs, ys)
    e0.record()
    F.copy_probe(xs, ys)
    e1.record()
    torch.cuda.synchronize()
    print(f"# {a.graph} {ns} streams x {T} samples, {'stream-major' if a.sm else 'tile ' + str(a.tile)}; B_alg = {b_alg / 1e9:.3f} GB; copy probe "
          f"{2 * n * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9:.0f} GB/s")
    for s, _ in vs:
        ts = sorted(times[s])
        med, mn = ts[len(ts) // 2], ts[0]
        print(json.dumps({"variant": s, "ms_med": round(med, 3), "ms_min": round(mn, 3),
                          "GBs_med": round(b_alg / med / 1e6, 1), "Msamples_s": round(ns * T / med / 1e3, 1),
                          "frac_8TBs": round(b_alg / med / 1e6 / 8000, 4)}))


if __name__ == "__main__":
    main()
