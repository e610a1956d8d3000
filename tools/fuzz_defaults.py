#!/usr/bin/env python3
"""Dev tool (GPU box): the library's DEFAULT choice (no variant) at random shapes -- stream counts from 64 to 1.3 M (powers of two, multiples of tiles, odd
counts), block lengths 1 .. 1500, plain rows / tiles / stream-major buffers -- for the BASELINE graphs and their neighbours: the whole output and the state
against a plain explicit variant (one stream per lane, 8-row chunks, no stage packing), sampled streams against the oracle.  Exercises resolve_variant's rules
(wave splits, I/O waves, stage packing, lockstep on rows / tiles / LDS rings, laps, remainders, ragged counts, stream-major bodies) rather than single kernels.
usage: tools/fuzz_defaults.py <first_seed> <count> [time limit in seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import graphs as G  # noqa: E402
from oracle import flowz_oracle as O  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

GRAPHS = {"cascade6": G.df1_cascade(6), "cascade4": G.df1_cascade(4), "cascade8": G.df1_cascade(8), "df1": G.df1(), "par4f": G.par4_sum_fanout(), "par4": G.par4_sum(),
          "lds_ring": G.lds_ring_comb(), "clipped_biquad": G.clipped_biquad(), "cascade6_gain": G.seq(G.df1_cascade(6), G.mul(G.lit(0.7), G.IN(1))), "osc": G.osc_chain(6)}
progs = {k: F.compile(F.from_sexpr(g)) for k, g in GRAPHS.items()}
first, count = int(sys.argv[1]), int(sys.argv[2])
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t_start, ok, bad, seen = time.time(), 0, 0, {}
for seed in range(first, first + count):
    if time.time() - t_start > limit:
        break
    rng = np.random.default_rng(seed)
    name = str(rng.choice(list(GRAPHS)))
    g, p = GRAPHS[name], progs[name]
    r = rng.random()
    ns = int(2 ** rng.uniform(6, 20.3))
    if r < 0.3:
        ns = 1 << int(rng.integers(6, 21))
    elif r < 0.5:
        ns = int(rng.choice([32768, 32769, 40000, 49152, 65536, 65537, 131072, 262144, 262145, 300000, 524288, 786432, 1000000, 1048576, 1048577]))
    T = int(rng.choice([1, 3, 64, 200, 256, 300, 1024, 1100, 1500])) if rng.random() < 0.7 else int(rng.integers(1, 1500))
    if ns * T * max(p.n_in, 1) > 3e9:                    # (keep a case below ~12 GiB of input frames)
        T = max(1, int(3e9 / (ns * max(p.n_in, 1))))
    nw = max(p.n_in, 1)
    layout = str(rng.choice(["rows", "tiles", "stream_major"]))
    tile = int(rng.choice([1024, 4096, 8192]))
    if layout == "tiles" and (ns % tile or ns <= tile):
        layout = "rows"
    if layout == "stream_major" and ((T * nw) % 4 or (T * p.n_out) % 4 or p.n_lds_slots and False):
        layout = "rows"
    params = None
    if p.n_param:
        import workloads as W
        params = torch.from_numpy(W.osc_chain_params(seed, np.arange(ns))).cuda()
    x = torch.empty((T, ns, nw), dtype=torch.float32, device="cuda")
    if name == "osc":
        x.zero_(); x[0].fill_(1.0)
    else:
        F.synth_fill(x, seed)
    plain = F.make_variant(1, 8, 256, F.C.FZ_VF_NO_STAGE_PACK)
    yr, sr = p.run_block(x, params=params, variant=plain)
    try:
        if layout == "rows":
            y, st = p.run_block(x, params=params)
            kn = p.kernel_name(None, ns, T)
        elif layout == "tiles":
            yt, st = p.run_block(F.to_tiled(x, tile), params=params)
            y = F.from_tiled(yt)
            kn = p.kernel_name(None, ns, T, tile)
        else:
            ys, st = p.run_block_stream_major(x.permute(1, 0, 2).contiguous(), params=params)
            y = ys.permute(1, 0, 2).contiguous()
            kn = p.kernel_name(F.make_variant(0, 0, 0, F.C.FZ_VF_STREAM_MAJOR), ns, T)
    except F.FlowzError as e:
        print(f"seed {seed}: {name} ns={ns} T={T} {layout}: REFUSED {e}", flush=True)
        bad += 1
        continue
    same = torch.equal(y.view(torch.int32), yr.view(torch.int32)) and torch.equal(st.view(torch.int32), sr.view(torch.int32))
    ids = np.unique(np.concatenate([[0, ns - 1], rng.integers(0, ns, 6)]))
    Tc = min(T, 300)                                      # (the Python oracle: a prefix of the block is enough to pin the reference kernel itself)
    xi = x[:Tc, torch.as_tensor(ids, device="cuda")].cpu().numpy()
    kw = {"params": params[:, torch.as_tensor(ids, device="cuda")].cpu().numpy()} if params is not None else {}
    want = O.compile(g, len(ids), **kw).run(xi)
    got = yr[:Tc, torch.as_tensor(ids, device="cuda")].cpu().numpy()
    same = same and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    seen[kn.replace("fz_block_kernel_", "")] = seen.get(kn.replace("fz_block_kernel_", ""), 0) + 1
    if same:
        ok += 1
    else:
        bad += 1
        print(f"seed {seed}: MISMATCH {name} ns={ns} T={T} {layout} tile={tile} kernel {kn}", flush=True)
    del x, y, yr
    torch.cuda.empty_cache()
print(f"default-plan fuzz seeds {first}..{seed}: {ok} shapes identical, {bad} mismatching or refused, {time.time() - t_start:.0f} s; kernels the defaults resolved to:")
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"   {n:4d}  {k}")
