#!/usr/bin/env python3
"""Round 6 (GPU box): config 2 (65 536 streams x 4096, plain time-major rows) -- board power, shader clock and joules per launch (rocm-smi, median
over 3 s of back-to-back launches) of the three arrangements the floor argument compares: the stage-packed single wave per SIMD, the wave split
W = 2 with both waves of a tuple on the SAME SIMD (four tuples per workgroup), and one compute wave + two I/O waves; next to a copy of the bytes."""
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
        time.sleep(0.25)


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
    w = med([s[1] for s in mine])
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "frac": round(bytes_ / ms / 1e6 / 8000, 4), "package_W": w, "sclk_MHz": med([s[2] for s in mine]),
                      "J_per_launch": round(w * ms / 1e3, 4) if w else None, "samples": len(mine)}), flush=True)


threading.Thread(target=sampler, daemon=True).start()
T, ns = 4096, 65536
prog = F.compile(F.from_sexpr(W.df1_cascade(6)))
C = F.C
x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
F.synth_fill(x, 1)
st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
b = ns * (8 * T + 8 * prog.n_state)
V = {"stage-packed single wave (default)": None,
     "W=2 split, same SIMD (1,16,256,1024)": (1, 16, 256, C.FZ_VF_WAVE_SPLIT),
     "W=2 split, same SIMD, rounds of 32 (1,32,256,1024)": (1, 32, 256, C.FZ_VF_WAVE_SPLIT),
     "one compute + two I/O waves": (1, 16, 0, C.FZ_VF_IO_WAVE | C.FZ_VF_IO_WAVE2),
     "W=3 split (1,16,256,2048)": (1, 16, 256, C.FZ_VF_WAVES(3))}
for rep in range(2):
    for name, v in V.items():
        vv = F.make_variant(*v) if v else None
        loop(f"{name} {prog.kernel_name(vv, ns, T)}", lambda: prog.run_block(x, state=st, out=y, variant=vv), b)
    loop("copy of the same bytes", lambda: F.copy_probe(x, y), 8 * ns * T)
stop = True
