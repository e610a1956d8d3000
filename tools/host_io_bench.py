#!/usr/bin/env python3
"""Dev tool (GPU box): PCIe-inclusive rate of the host-frames path (fz_bank_process_host) on the
6-biquad cascade: pipelined (pinned / pageable host memory) vs one synchronous round trip."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from zignal_amd import workloads as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

ns, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
bank = prog.bank(ns)
xd = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
F.synth_fill(xd, 20160512)
gb = 2 * T * ns * 4 / 1e9
for name, pinned in (("pinned", True), ("pageable", False)):
    x = torch.empty((T, ns, 1), dtype=torch.float32, pin_memory=pinned)
    x.copy_(xd)
    out = torch.empty((T, ns, 1), dtype=torch.float32, pin_memory=pinned)
    ts = []
    for _ in range(4):
        bank.reset()
        t0 = time.perf_counter()
        bank.process_host(x, out=out)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"pipelined, {name:8s} host memory: {t * 1e3:8.2f} ms  {ns * T / t / 1e6:9.1f} Msamples/s  {gb / t:6.1f} GB/s over PCIe (both directions)")
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        d = x.cuda()
        y, _ = prog.run_block(d)
        out.copy_(y)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"sequential H2D, kernel, D2H, {name:8s}: {t * 1e3:8.2f} ms  {ns * T / t / 1e6:9.1f} Msamples/s  {gb / t:6.1f} GB/s")

# the reference's own calling convention: one contiguous host buffer per stream
for name, pinned in (("pinned", True), ("pageable", False)):
    xs = torch.empty((ns, T, 1), dtype=torch.float32, pin_memory=pinned)
    xs.copy_(xd.permute(1, 0, 2))
    outs = torch.empty((ns, T, 1), dtype=torch.float32, pin_memory=pinned)
    ts = []
    for _ in range(4):
        bank.reset()
        t0 = time.perf_counter()
        bank.process_host_stream_major(xs, out=outs)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"stream-major host rows, {name:8s}: {t * 1e3:8.2f} ms  {ns * T / t / 1e6:9.1f} Msamples/s  {gb / t:6.1f} GB/s over PCIe (both directions)")
