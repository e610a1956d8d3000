"""Randomised graphs: lowering vs oracle on the CPU (IR interpreter, tests only) and the fused
kernels vs the oracle on the GPU."""
import numpy as np
import pytest

import randgraphs as R
from ir_interp import run_ir
from oracle import flowz_oracle as O
from zignal_amd import flowz as F


def same_or_both_nan(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(nan, 0, a).view(np.uint32), np.where(nan, 0, b).view(np.uint32))


def usable(seed):
    g, n_in, n_out = R.make(seed)
    try:
        ok = O.input_arity(g) == n_in and O.output_arity(g) == n_out
        O.compile(g, 1)
    except O.GraphError:
        return None
    return (g, n_in, n_out) if ok else None


def test_generator_yields_mostly_valid_graphs():
    assert sum(usable(s) is not None for s in range(200)) > 150


@pytest.mark.parametrize("chunk", range(8))
def test_lowering_matches_oracle_on_random_graphs(chunk):
    n = 0
    for seed in range(chunk * 25, chunk * 25 + 25):
        u = usable(seed)
        if u is None:
            continue
        g, n_in, n_out = u
        p = F.compile(F.from_sexpr(g))
        assert (p.n_in, p.n_out) == (n_in, n_out)
        x = O.synth_input(seed, np.arange(2), 40, n_wires=n_in)
        want = O.compile(g, 2).run(x)
        got, _ = run_ir(p, x)
        assert same_or_both_nan(got, want), f"seed {seed}: {g}"
        assert np.isfinite(want).all(), f"seed {seed} blew up"
        n += 1
    assert n >= 15


def usable_cmp(seed):
    g, n_in, n_out = R.make_cmp(seed)
    try:
        ok = O.input_arity(g) == n_in and O.output_arity(g) == n_out
        O.compile(g, 1)
    except O.GraphError:
        return None
    return (g, n_in, n_out) if ok and any(k in str(g) for k in ("'lt'", "'ge'", "'and'", "'not'", "'ne'", "'eq'", "'le'", "'gt'", "'or'")) else None


@pytest.mark.parametrize("chunk", range(4))
def test_lowering_matches_oracle_on_random_graphs_with_comparisons(chunk):
    """the fuzz family of SURVEY 8 row a4's widening: comparison and logical operators sprinkled over the random graphs"""
    n = 0
    for seed in range(6000 + chunk * 25, 6000 + chunk * 25 + 25):
        u = usable_cmp(seed)
        if u is None:
            continue
        g, n_in, n_out = u
        p = F.compile(F.from_sexpr(g))
        assert (p.n_in, p.n_out) == (n_in, n_out) and not p.stage_packable
        x = O.synth_input(seed, np.arange(3), 40, n_wires=n_in)
        want = O.compile(g, 3).run(x)
        got, _ = run_ir(p, x)
        assert same_or_both_nan(got, want), f"seed {seed}: {g}"
        assert np.isfinite(want).all(), f"seed {seed} blew up"
        n += 1
    assert n >= 12


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(3))
def test_kernels_match_oracle_on_random_graphs_with_comparisons(chunk):
    import torch
    ns, T = 136, 44
    n = 0
    for seed in range(7000 + chunk * 20, 7000 + chunk * 20 + 20):
        u = usable_cmp(seed)
        if u is None:
            continue
        g, n_in, n_out = u
        p = F.compile(F.from_sexpr(g))
        x = O.synth_input(seed, np.arange(ns), T, n_wires=n_in)
        want = O.compile(g, ns).run(x)
        xd = torch.from_numpy(x).cuda()
        for P in (1, 2, 4):
            y, _ = p.run_block(xd, variant=F.make_variant(P, 8))
            assert same_or_both_nan(y.cpu().numpy(), want), f"seed {seed} P={P}: {g}"
        ya, st = p.run_block(xd[:19].contiguous())
        yb, st = p.run_block(xd[19:].contiguous(), state=st)
        assert same_or_both_nan(torch.cat([ya, yb]).cpu().numpy(), want), f"seed {seed} chained"
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        if (T * n_in) % 4 == 0 and (T * n_out) % 4 == 0:
            ys, _ = p.run_block_stream_major(xs)
            assert same_or_both_nan(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want), f"seed {seed} stream-major"
        n += 1
    assert n >= 10


def same64(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(np.where(nan, 0, a).view(np.uint64), np.where(nan, 0, b).view(np.uint64))


def usable_typed(seed):
    g, n_in, n_out, kind = R.make_typed(seed)
    try:
        if O.input_arity(g) != n_in or O.output_arity(g) != n_out:
            return None
        O.compile(g, 1)
    except O.GraphError:
        return None
    return g, n_in, n_out, kind


@pytest.mark.parametrize("chunk", range(4))
def test_lowering_matches_oracle_on_random_typed_graphs(chunk):
    """double literals and std::complex stages sprinkled over the random graphs"""
    kinds = set()
    n = 0
    for seed in range(3000 + chunk * 25, 3000 + chunk * 25 + 25):
        u = usable_typed(seed)
        if u is None:
            continue
        g, n_in, n_out, kind = u
        p = F.compile(F.from_sexpr(g))
        f = O.compile(g, 2)
        assert (p.n_in, p.info.n_out_wires, p.n_out) == (n_in, n_out, f.n_slots)
        assert p.output_dtypes() == f.out_types
        x = O.synth_input(seed, np.arange(2), 40, n_wires=n_in)
        got, _ = run_ir(p, x)
        assert same_or_both_nan(got, f.run(x)), f"seed {seed}: {g}"
        kinds.add(kind if kind in ("complex", "cdouble") else ("double" if p.n_const64 else "float"))
        n += 1
    assert n >= 12 and {"complex", "double"} <= kinds


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(3))
def test_kernels_match_oracle_on_random_typed_graphs(chunk):
    import torch
    ns, T = 136, 45
    n = 0
    for seed in range(4000 + chunk * 20, 4000 + chunk * 20 + 20):
        u = usable_typed(seed)
        if u is None:
            continue
        g, n_in, n_out, kind = u
        p = F.compile(F.from_sexpr(g))
        x = O.synth_input(seed, np.arange(ns), T, n_wires=n_in)
        want = O.compile(g, ns).run(x)
        want64 = O.compile(g, ns, out_f64=True).run(x)
        xd = torch.from_numpy(x).cuda()
        for P in (1, 2, 4):
            y, _ = p.run_block(xd, variant=F.make_variant(P, 8))
            assert same_or_both_nan(y.cpu().numpy(), want), f"seed {seed} P={P}: {g}"
        y64, _ = p.run_block(xd, out_f64=True)
        assert same64(y64.cpu().numpy(), want64), f"seed {seed} f64 frames: {g}"
        ya, st = p.run_block(xd[:19].contiguous())
        yb, st = p.run_block(xd[19:].contiguous(), state=st)
        assert same_or_both_nan(torch.cat([ya, yb]).cpu().numpy(), want), f"seed {seed} chained"
        n += 1
    assert n >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(6))
def test_kernels_match_oracle_on_random_graphs(chunk):
    import torch
    ns, T = 136, 45
    n = 0
    for seed in range(1000 + chunk * 20, 1000 + chunk * 20 + 20):
        u = usable(seed)
        if u is None:
            continue
        g, n_in, n_out = u
        p = F.compile(F.from_sexpr(g))
        x = O.synth_input(seed, np.arange(ns), T, n_wires=n_in)
        want = O.compile(g, ns).run(x)
        xd = torch.from_numpy(x).cuda()
        for P in (1, 2, 4):
            y, _ = p.run_block(xd, variant=F.make_variant(P, 8))
            assert same_or_both_nan(y.cpu().numpy(), want), f"seed {seed} P={P}: {g}"
        # split into two chained blocks at an odd point
        ya, st = p.run_block(xd[:19].contiguous())
        yb, st = p.run_block(xd[19:].contiguous(), state=st)
        assert same_or_both_nan(torch.cat([ya, yb]).cpu().numpy(), want), f"seed {seed} chained"
        n += 1
    assert n >= 10
