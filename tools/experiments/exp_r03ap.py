#!/usr/bin/env python3
"""Experiment (GPU box): power / clock / time of the cascade kernels with their frame traffic sent through zero-byte descriptors
(FLOWZ_HIP_EXTRA_OPTS="-DFZ_DBG_NOLOAD -DFZ_DBG_NOSTORE"): what the arithmetic (+ LDS transposition) costs without HBM."""
import os, sys, subprocess, threading, time, json, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zignal_amd import workloads as G, flowz as F
rows, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            c = next(iter(json.loads(o).values()))
            rows.append((time.time(), float([v for k, v in c.items() if "Power (W)" in k][0]), int(re.search(r"(\d+)", c["sclk clock speed:"]).group(1))))
        except Exception:
            pass
        time.sleep(0.2)
threading.Thread(target=sampler, daemon=True).start()
ns, T = 1 << 20, 4096
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
st = torch.zeros((prog.n_state, ns), device="cuda")
xt, ot = x.view(T, ns, 1), out.view(T, ns, 1)
LG = F.C.FZ_VF_LOCKSTEP | F.C.FZ_VF_GRID_SYNC
cases = [("stream-major U=128", lambda: prog.run_block_stream_major(x, state=st, out=out, variant=F.make_variant(1, 128, 0, 256))),
         ("time-major p4u1 lockstep|sync", lambda: prog.run_block(xt, state=st, out=ot, variant=F.make_variant(4, 1, 1024, LG | F.C.FZ_VF_PREFETCH3)))]
for name, fn in cases:
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.5:
        for _ in range(20): fn()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    mine = [r for r in rows if r[0] > t0 + 1.0]
    ms = e0.elapsed_time(e1) / n
    w = sorted(r[1] for r in mine)[len(mine) // 2]
    print(f"opts='{os.environ.get('FLOWZ_HIP_EXTRA_OPTS', '')}' {name:32s} {ms:7.3f} ms/launch, {w:6.0f} W, sclk {sorted(r[2] for r in mine)[len(mine) // 2]} MHz, {w * ms / 1e3:5.2f} J/launch", flush=True)
stop = True
