#!/usr/bin/env python3
"""Experiment (GPU box): board power and clocks (rocm-smi, sampled every 0.25 s) while one kernel runs in a loop for a few seconds."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e)))
        time.sleep(0.25)


def loop(name, fn, seconds=4.0):
    global samples
    fn(); torch.cuda.synchronize()
    samples = []
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
    mine = [s for s in samples if s[0] > t0 + 1.0]
    print(f"## {name}: {ms:.3f} ms per launch over {n} launches; {len(mine)} samples after the first second")
    for _, o in mine[:: max(1, len(mine) // 4)]:
        print("   ", " ".join(o.split())[:600])
    sys.stdout.flush()


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.5)
print("## idle:", " ".join(samples[-1][1].split())[:600] if samples else None)
ns, T = 1 << 20, 4096
x = torch.randn((ns, T, 1), device="cuda") * 0.1
out = torch.empty((ns, T, 1), device="cuda")
for name, mk in (("cascade6", lambda: G.df1_cascade(6)), ("df1", G.df1)):
    prog = F.compile(F.from_sexpr(mk()))
    st = torch.zeros((prog.n_state, ns), device="cuda")
    v = F.make_variant(1, 128, 0, 256)
    loop(f"stream-major {name} U=128", lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v))
    if name == "cascade6":
        v64 = F.make_variant(1, 64, 0, 256)
        loop(f"stream-major {name} U=64", lambda: prog.run_block_stream_major(x, state=st, out=out, variant=v64))
    xt, ot = x.view(T, ns, 1), out.view(T, ns, 1)
    loop(f"time-major {name} (library default)", lambda: prog.run_block(xt, state=st, out=ot))
loop("copy kernel", lambda: F.copy_probe(x, out))
stop = True
