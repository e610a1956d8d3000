"""Per-stream coefficient generators for the BASELINE workloads, numpy only."""
import numpy as np

F32 = np.float32


def _unit01(seed, streams, j):
    """hash -> [0,1) float64, deterministic in (seed, stream, j)."""
    from oracle.flowz_oracle import hash32
    h = hash32(seed, np.asarray(streams, np.uint64), np.uint64(j))
    return h.astype(np.float64) / 4294967296.0


def osc_chain_params(seed, streams, n_stage=6):
    """[1+5n, n_streams] float32: k = 2cos(theta), theta in (0.05,3.0); per stage r in [0.5,0.95],
    a1 = 2 r cos(phi), a2 = -r^2, b in [-0.5,0.5]  (SURVEY 8d config 4)."""
    streams = np.asarray(streams, np.uint64)
    P = np.empty((1 + 5 * n_stage, len(streams)), F32)
    theta = 0.05 + 2.95 * _unit01(seed, streams, 0)
    P[0] = (2.0 * np.cos(theta)).astype(F32)
    for j in range(n_stage):
        base = 1 + 5 * j
        for q in range(3):
            P[base + q] = (_unit01(seed, streams, 10 * (j + 1) + q) - 0.5).astype(F32)
        r = 0.5 + 0.45 * _unit01(seed, streams, 10 * (j + 1) + 3)
        phi = np.pi * _unit01(seed, streams, 10 * (j + 1) + 4)
        P[base + 3] = (2.0 * r * np.cos(phi)).astype(F32)
        P[base + 4] = (-(r * r)).astype(F32)
    return P
