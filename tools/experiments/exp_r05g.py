#!/usr/bin/env python3
"""Round 5 (GPU box): is config 2 (65 536 streams x 4096, one wave per SIMD) at the board's power cap?  Board power and shader clock (rocm-smi)
while the stage-packed single wave, the two-I/O-wave kernel and a copy of the same bytes run back to back for 4 s each."""
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

from zignal_amd import flowz as F, workloads as W  # noqa: E402

samples, stop = [], False


def sampler():
    while not stop:
        try:
            o = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = next(iter(o.values()))
            w = [float(v) for k, v in c.items() if "Power (W)" in k and "Max" not in k]
            s = [int(re.search(r"(\d+)", v).group(1)) for k, v in c.items() if k.startswith("sclk clock speed")]
            samples.append((time.time(), w[0] if w else None, s[0] if s else None))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.3)


def loop(name, fn, bytes_, seconds=4.0):
    fn(); torch.cuda.synchronize()
    t0, n = time.time(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(200):
            fn()
        n += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    mine = [s for s in samples if s[0] > t0 + 1.0 and s[1]]
    med = lambda v: sorted(v)[len(v) // 2] if v else None  # noqa: E731
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "frac": round(bytes_ / ms / 1e6 / 8000, 4), "package_W": med([s[1] for s in mine]), "sclk_MHz": med([s[2] for s in mine]), "samples": len(mine)}), flush=True)


threading.Thread(target=sampler, daemon=True).start()
os.environ["FLOWZ_HIP_AUTOTUNE"] = "0"
T = 4096
prog = F.compile(F.from_sexpr(W.df1_cascade(6)))
for ns in (65536, 32768, 16384, 1 << 20):
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    F.synth_fill(x, 1)
    st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
    b = ns * (8 * T + 8 * prog.n_state)
    loop(f"{ns} default " + prog.kernel_name(None, ns, T), lambda: prog.run_block(x, state=st, out=y), b)
    if ns == 65536:
        v = F.make_variant(1, 16, 0, F.C.FZ_VF_IO_WAVE | F.C.FZ_VF_IO_WAVE2)
        loop(f"{ns} two I/O waves", lambda: prog.run_block(x, state=st, out=y, variant=v), b)
    loop(f"{ns} copy", lambda: F.copy_probe(x, y), 8 * ns * T)
    del x, y, st
stop = True
