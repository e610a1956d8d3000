#!/usr/bin/env python3
"""Round 5 (GPU box): what do LAPS cost?  The 4-wire sum on plain time-major rows.  (a) 262 144 streams, one lap, rows 4 MiB apart and
wholly consumed: the whole block is ONE sequential sweep through memory; (b) 1 048 576 streams, four laps: each lap takes a contiguous
quarter (4 MiB) of rows that lie 16 MiB apart; (c) ONE of those four laps alone (FLOWZ_HIP_ONLY_LAP, set by the caller of this script):
the same work as (a) on the strided rows.  If (c) takes what (a) takes, the laps' cost is in their succession; if it takes a quarter of
(b), it is the stride."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F, workloads as W  # noqa: E402

T = 4096
prog = F.compile(F.from_sexpr(W.par4_sum()))
os.environ["FLOWZ_HIP_AUTOTUNE"] = "0"
V = F.make_variant(*[int(v) for v in os.environ["FZ_VARIANT"].split(",")]) if os.environ.get("FZ_VARIANT") else None   # (the same kernel on both shapes)
for ns in [int(a) for a in sys.argv[1:]] or [262144, 1 << 20]:
    x = torch.empty((T, ns, 4), dtype=torch.float32, device="cuda")
    y = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, 20160512)
    st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
    for _ in range(2):
        prog.run_block(x, state=st, out=y, variant=V)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        prog.run_block(x, state=st, out=y, variant=V)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"streams": ns, "only_lap": os.environ.get("FLOWZ_HIP_ONLY_LAP"), "kernel": prog.kernel_name(V, ns, T), "ms": round(ms, 3),
                      "frac_if_whole_block": round(ns * (4 * T * 5 + 8 * prog.n_state) / ms / 1e6 / 8000, 4)}), flush=True)
    del x, y, st
