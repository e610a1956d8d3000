#!/usr/bin/env python3
"""Dev tool (GPU box): rate of the stream-major <-> frames adapter (fz_transpose_frames)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zignal_amd import flowz as F  # noqa: E402

for ns, T, w, tile in ((1 << 20, 1024, 1, 8192), (1 << 20, 1024, 1, 0), (65536, 4096, 1, 8192), (1 << 18, 1024, 4, 4096)):
    x = torch.randn((ns, T, w), device="cuda")
    fr = F.frames_from_stream_major(x, tile)
    back = F.frames_to_stream_major(fr)
    torch.cuda.synchronize()
    res = []
    for fn in (lambda: F.frames_from_stream_major(x, tile, out=fr), lambda: F.frames_to_stream_major(fr, out=back)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(2 * x.numel() * 4 / (e0.elapsed_time(e1) / 5 / 1e3) / 1e9)
    print(f"{ns} streams x {T} samples x {w} wires, tile {tile}: to frames {res[0]:7.1f} GB/s, to stream-major {res[1]:7.1f} GB/s (read + write)")
