#!/usr/bin/env python3
"""Experiment (GPU box): shader clock and power (rocm-smi) while the FEW-STREAM kernels run back to back for 3 s each."""
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
        except Exception as e:
            pass
        time.sleep(0.2)
threading.Thread(target=sampler, daemon=True).start()
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
T = 4096
for ns in (8192, 16384, 32768, 65536, 1 << 20):
    tile = 8192 if ns > 8192 else 0
    x = torch.randn((ns // tile, T, tile, 1) if tile else (T, ns, 1), device="cuda") * 0.1
    y = torch.empty_like(x)
    st = torch.zeros((prog.n_state, ns), device="cuda")
    fn = lambda: prog.run_block(x, state=st, out=y)
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(200): fn()
        n += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    mine = [r for r in rows if r[0] > t0 + 1.0]
    print(f"{ns:8d} streams: {prog.kernel_name(None, ns, T, tile):40s} {e0.elapsed_time(e1) / n:8.4f} ms/launch over {n} launches; "
          f"power {min(r[1] for r in mine):.0f}-{max(r[1] for r in mine):.0f} W, sclk {min(r[2] for r in mine)}-{max(r[2] for r in mine)} MHz", flush=True)
stop = True
