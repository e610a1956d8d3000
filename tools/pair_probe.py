#!/usr/bin/env python3
"""Dev tool (GPU box): the PAIR long-run stream-major body (two streams per lane, FZ_VF_SM_LONG) against the library's default
stream-major kernel -- bit-for-bit on ragged shapes, then the time at 1 M streams x 4096 samples."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

PAIR = F.make_variant(2, 64, 0, 256)
graphs = {"cascade6": lambda: G.df1_cascade(6), "cascade2": lambda: G.df1_cascade(2), "df1": G.df1}
QUICK = bool(os.environ.get("PAIR_PROBE_QUICK"))          # the cascade only: parity, then the time at 1 M streams x 4096
bad = 0
for name in (["cascade6"] if QUICK else graphs):
    prog = F.compile(F.from_sexpr(graphs[name]()))
    for ns, T in ((2, 128), (130, 256), (776, 300), (1024, 124), (4098, 1000), (70000, 516)):
        torch.manual_seed(ns + T)
        x = torch.randn((ns, T, 1), device="cuda") * 0.1
        y0, s0 = prog.run_block_stream_major(x)
        y1, s1 = prog.run_block_stream_major(x, variant=PAIR)
        # chained: two windows, state carried
        cut = (T // 2) // 4 * 4
        out = torch.zeros_like(y0)
        _, s2 = prog.run_block_stream_major(x, out=out, n_samples=cut, variant=PAIR)
        prog.run_block_stream_major(x, out=out, state=s2, row0=cut, variant=PAIR)
        ok = torch.equal(y0.view(torch.int32), y1.view(torch.int32)) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)) and \
            torch.equal(out.view(torch.int32), y0.view(torch.int32)) and torch.equal(s2.view(torch.int32), s0.view(torch.int32))
        bad += not ok
        print(f"{name:9s} {ns:6d} x {T:5d}: {'identical' if ok else 'MISMATCH'}", flush=True)
print("parity:", "all identical" if not bad else f"{bad} MISMATCHES", flush=True)


def timed(fn, reps=8):
    fn()
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


os.environ["FLOWZ_HIP_AUTOTUNE"] = "0"
for name, ns, T in ((("cascade6", 1 << 20, 4096),) if QUICK else
                    (("cascade6", 1 << 20, 4096), ("cascade2", 1 << 20, 4096), ("df1", 1 << 20, 4096), ("cascade6", 1 << 18, 4096), ("cascade6", 1 << 20, 1024))):
    prog = F.compile(F.from_sexpr(graphs[name]()))
    x = torch.randn((ns, T, 1), device="cuda") * 0.1
    out = torch.empty((ns, T, 1), device="cuda")
    st = torch.zeros((prog.n_state, ns), device="cuda")
    b = ns * T * 8
    print(f"# {name}, {ns} streams x {T} samples, B_alg {b / 1e9:.2f} GB")
    for rnd in range(2):
        more = [("pair, 128-lane workgroups", F.make_variant(2, 64, 128, 256)), ("pair, 64-lane workgroups", F.make_variant(2, 64, 64, 256))] if os.environ.get("PAIR_PROBE_BLOCKS") else []
        for label, v in [("default", None), ("one stream per lane", F.make_variant(1, 128, 0, 256)), ("pair P=2 U=64 SM_LONG", PAIR)] + more:
            ms = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
            kv = F.make_variant(0, 0, 0, 128) if v is None else F.make_variant(v.streams_per_lane, v.unroll, v.block_threads, v.flags | 128)
            print(f"  {label:24s} {prog.kernel_name(kv, ns, T, 0):40s} {ms:8.3f} ms  {b / ms / 1e6:7.1f} GB/s  frac {b / ms / 1e6 / 8000:.4f}", flush=True)
    del x, out, st
    torch.cuda.empty_cache()
