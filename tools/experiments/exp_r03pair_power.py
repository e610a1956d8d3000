#!/usr/bin/env python3
"""Experiment (GPU box): board power and shader clock (rocm-smi, every 0.25 s) while the stream-major cascade runs in a loop for 4 s --
the pair long-run body against the one-stream body, the time-major kernel next to them."""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

samples, stop = [], False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e)))
        time.sleep(0.25)


def loop(name, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    W, clk = [], []
    for t, o in samples:
        if t < t0 + 1.0:
            continue
        try:
            card = next(iter(json.loads(o).values()))
        except Exception:  # noqa: BLE001
            continue
        for k, v in card.items():
            if "Package Power" in k and "Max" not in k:
                W.append(float(v))
            if k.startswith("sclk"):
                m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                if m:
                    clk.append(int(m.group(1)))
    w = sorted(W)[len(W) // 2] if W else float("nan")
    c = sorted(clk)[len(clk) // 2] if clk else 0
    print(f"{name:46s} {ms:7.3f} ms per launch over {n:4d} launches   {w:7.1f} W   sclk {c:5d} MHz   {w * ms / 1e3:6.2f} J per launch", flush=True)


threading.Thread(target=sampler, daemon=True).start()
time.sleep(1.0)
ns, T = 1 << 20, 4096
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
st = torch.zeros((prog.n_state, ns), device="cuda")
for rnd in range(2):
    loop("stream-major, pair body (default)", lambda: prog.run_block_stream_major(x, state=st, out=out))
    v1 = F.make_variant(1, 128, 0, 256)
    loop("stream-major, one stream per lane (stage-packed)", lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v1))
    xt, ot = x.view(T, ns, 1), out.view(T, ns, 1)
    loop("time-major (library default)", lambda: prog.run_block(xt, state=st, out=ot))
stop = True
