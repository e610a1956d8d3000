#!/usr/bin/env python3
"""Example: a 6-stage biquad cascade over 1 M streams with device-resident, stream-tiled frames."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import flowz as F          # noqa: E402
from zignal_amd.flowz import _1, _2        # noqa: E402

b0, b1, b2, a1, a2 = 0.05, -0.075, 0.275, 0.2, -0.8
stage = (b0 * _1 + b1 * _1[_1] + b2 * _1[_2]) >> ~(_2 + a1 * _1[_1] + a2 * _1[_2])      # fwd |= bwd
prog = F.compile(F.seq(*[stage] * 6))
print({k: getattr(prog, k) for k in ("n_in", "n_out", "n_ops", "n_state", "stage_packable")})

n_streams, n_samples = 1 << 20, 1024
tile = prog.recommended_tile_streams()
x = torch.empty((n_streams // tile, n_samples, tile, 1), device="cuda")
F.synth_fill(x, seed=1)
plan, ms = prog.tune(x)                                        # optional: measure the kernel variants on this board once
print(f"plan: {plan.streams_per_lane} streams/lane, unroll {plan.unroll}, flags {plan.flags:#x}: {ms:.3f} ms per block")
y, state = prog.run_block(x)                                   # one launch; state carries to the next block
y2, state = prog.run_block(x, state=state)
torch.cuda.synchronize()
print("kernel:", prog.kernel_name(plan, n_streams, n_samples), "| out", tuple(y.shape), "| finite:", bool(torch.isfinite(y2).all()))
