#!/usr/bin/env python3
"""GPU box, round 4: does the position of `out` relative to `in` matter on plain time-major frames whose row pitch is a large power of two?
(in-row t+2, in-row t+1 and out-row t of a lap are 2^k bytes apart when both buffers come from the allocator on the same 2^k grid)
usage: exp_r04t.py  -- prints ms per block for out shifted by 0 / 4 KiB / 68 KiB / 1 MiB + 4 KiB / 3 MiB + 12 KiB floats-bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F  # noqa: E402
from zignal_amd import workloads as G  # noqa: E402

os.environ.setdefault("FLOWZ_HIP_AUTOTUNE", "0")


def timed(fn, n=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for name, g, ns in (("cascade6", G.df1_cascade(6), 1 << 20), ("cascade6", G.df1_cascade(6), 1 << 21), ("cascade6", G.df1_cascade(6), 2_000_000),
                    ("par4", G.par4_sum(), 1 << 20), ("par4", G.par4_sum(), 1_040_000), ("par4", G.par4_sum(), 1 << 19)):
    prog = F.compile(F.from_sexpr(g))
    T = 4096
    x = torch.empty((T, ns, max(prog.n_in, 1)), dtype=torch.float32, device="cuda")
    F.synth_fill(x, 20160512)
    st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
    n_out = T * ns * prog.n_out
    pad = 8 << 20
    ybuf = torch.empty(n_out + pad, dtype=torch.float32, device="cuda")
    b_alg = ns * (4 * T * (prog.n_in + prog.n_out) + 8 * prog.n_state)
    print(f"# {name} {ns} streams x {T}: in row pitch {ns * prog.n_in * 4} B, out row pitch {ns * prog.n_out * 4} B; x at {x.data_ptr():#x}, ybuf at {ybuf.data_ptr():#x}; kernel {prog.kernel_name(None, ns, T)}")
    for shift in (0, 4096, 69632, (1 << 20) + 4096, (3 << 20) + 12288, (4 << 20), (6 << 20) + 65536):
        y = ybuf[shift // 4: shift // 4 + n_out].view(T, ns, prog.n_out)
        ms = timed(lambda: prog.run_block(x, state=st, out=y))
        print(f"   out shifted by {shift:9d} bytes: {ms:8.3f} ms  {b_alg / ms / 1e6 / 8000:.4f} of 8 TB/s")
    del x, ybuf, st
    torch.cuda.empty_cache()
