#!/usr/bin/env python3
"""Dev tool (GPU box): the stream-major bodies on graphs of growing arithmetic weight, store-policy variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zignal_amd import flowz as F, workloads as G

def timed(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

graphs = {"gain": lambda: G.mul(G.lit(0.5), G.IN(1)), "df1": G.df1, "cascade2": lambda: G.df1_cascade(2), "cascade6": lambda: G.df1_cascade(6)}
ns, T = 1 << 20, 2048
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
b = ns * T * 8
for name in sys.argv[1:] or list(graphs):
    prog = F.compile(F.from_sexpr(graphs[name]()))
    st = torch.zeros((max(prog.n_state, 1), ns), device="cuda")
    for label, v in (("long U=128", F.make_variant(1, 128, 0, 256)), ("long U=128 nopack", F.make_variant(1, 128, 0, 256 | 16)),
                     ("long U=128 st=nt", F.make_variant(1, 128, 0, 256 | (7 << 16))), ("long U=128 st=plain", F.make_variant(1, 128, 0, 256 | (1 << 16))),
                     ("long U=128 st=sc1", F.make_variant(1, 128, 0, 256 | (3 << 16))), ("long U=128 ld=plain", F.make_variant(1, 128, 0, 256 | (1 << 12))),
                     ("long U=128 nopack st=nt", F.make_variant(1, 128, 0, 256 | 16 | (7 << 16))),
                     ("short U=32", F.make_variant(0, 0, 0, 512)), ("short U=32 nopack", F.make_variant(1, 32, 0, 16))):
        try:
            ms = timed(lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
            print(f"{name:9s} {label:26s} {ms:8.3f} ms  {b / ms / 1e6:7.1f} GB/s  {prog.kernel_name(F.make_variant(v.streams_per_lane, v.unroll, v.block_threads, v.flags | 128), ns, T)}")
        except F.FlowzError as e:
            print(f"{name:9s} {label:26s} -- {str(e)[:80]}")
