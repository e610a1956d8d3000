"""Parity tests proper (MI355X): the HIP path, called through the C ABI, against the oracle.

Bar: BIT-EXACT float32 (0 ULP).  BASELINE.json's north_star allows 1 ULP; these tests assert 0."""
import ctypes
import json
import os

import numpy as np
import pytest

import graphs as G
import workloads as W
from oracle import coracle as C
from oracle import flowz_oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "tests_cpp_known_answers.json")))
REF = json.load(open(os.path.join(HERE, "golden", "ref_biquad_vectors.json")))
SEED = 20160512


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    torch.cuda.set_device(0)
    return torch


@pytest.fixture(scope="module")
def F():
    from zignal_amd import flowz
    assert flowz.device_count() >= 1
    return flowz


def tup(x):
    return tuple(tup(v) for v in x) if isinstance(x, list) else x


def bits(hexlist):
    return np.array([int(h, 16) for h in hexlist], np.uint32).view(np.float32)


def ndiff(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return int((a.view(np.uint32) != b.view(np.uint32)).sum())


def ndiff_nan_aware(a, b):
    """differing bit patterns, NaNs of any payload counted as equal to each other (a NaN's payload is not part of any contract)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())


def run_gpu(torch, F, prog, x_host, params=None, variant=None, state=None):
    x = torch.from_numpy(np.ascontiguousarray(x_host)).cuda()
    p = torch.from_numpy(np.ascontiguousarray(params)).cuda() if params is not None else None
    y, st = prog.run_block(x, state=state, params=p, variant=variant)
    torch.cuda.synchronize()
    return y.cpu().numpy(), st


# ---- the reference's own known answers, through the per-sample call protocol -------------------
@pytest.mark.parametrize("case", KA["evaluation"] + KA["readme"], ids=lambda c: c["name"])
def test_tests_cpp_known_answers_per_sample_calls(F, case):
    """compile(expr)(x...) one sample at a time (test/tests.cpp:88-178) via fz_bank_process_host."""
    from zignal_amd import _capi
    prog = F.compile(F.from_sexpr(tup(case["graph"])))
    bank = ctypes.c_void_p()
    _capi.check(_capi.lib.fz_bank_create(prog._h, 1, ctypes.byref(bank)))
    try:
        for ins, outs in case["calls"]:
            xi = (ctypes.c_float * len(ins))(*[float(v) for v in ins])
            yo = (ctypes.c_float * len(outs))()
            _capi.check(_capi.lib.fz_bank_process_host(bank, xi, yo, 1))
            assert [float(v) for v in yo] == [float(v) for v in outs]
    finally:
        _capi.lib.fz_bank_destroy(bank)


def test_graph_the_shipped_reference_misroutes_runs_per_the_arity_table(torch_cuda, F):
    """SURVEY App. C.1, the graph of test/tests.cpp:67-71: ~(_1 + _2[_1] |= _1[_1] + _2).  The reference asserts its arities and delays only;
    its shipped binary_feedback (flowz.hpp:1045-1050) would return 10, 30, 70 on (10,1),(20,2),(30,3).  The library routes per the arity
    table -- 1, 3, 16 -- says so in fz_info, and the kernels agree with the oracle on noise."""
    g = tup([c for c in KA["arity"] if c["name"] == "fb_two_inputs"][0]["graph"])
    prog = F.compile(F.from_sexpr(g))
    assert prog.differs_from_reference == 1 and prog.note.startswith("note:")
    x = np.array([[[10, 1]], [[20, 2]], [[30, 3]]], np.float32)
    y, _ = run_gpu(torch_cuda, F, prog, x)
    assert y.ravel().tolist() == [1.0, 3.0, 16.0]
    ns, T = 300, 257
    xn = O.synth_input(SEED + 21, np.arange(ns), T, n_wires=2)
    want = O.compile(g, ns).run(xn)
    for P in (1, 2, 4):
        got, _ = run_gpu(torch_cuda, F, prog, xn, variant=F.make_variant(P, 8))
        assert ndiff(got, want) == 0


# ---- golden vectors produced by the reference's own hand-written filters ------------------------
FORMS = {"df1": G.df1, "df2": G.df2, "df1t": G.df1t, "df1x2": lambda: G.seq(G.df1(), G.df1()),
         "df1x6": lambda: G.seq(*[G.df1() for _ in range(6)])}       # six reference DF1 closures in series: the headline workload's shape


@pytest.mark.parametrize("drive", ["dirac", "noise"])
def test_reference_golden_vectors_four_closures_side_by_side(torch_cuda, F, drive):
    """Config 3's shape against a composition of the reference's own DF1 closures (four instances, outputs summed left to right),
    every lane packing, and through stream-major buffers (the hold body: 201 / 1024 rows)."""
    g = G.seq(G.par(G.df1(), G.df1(), G.df1(), G.df1()), G.add(G.add(G.add(G.IN(1), G.IN(2)), G.IN(3)), G.IN(4)))
    x = np.ascontiguousarray(np.repeat(bits(REF["inputs4"][drive]).reshape(-1, 1, 4), 2, axis=1))      # the same four wires on two streams
    want = bits(REF["outputs4"][drive]["par4"])
    prog = F.compile(F.from_sexpr(g))
    for v in (None, F.make_variant(1, 8), F.make_variant(2, 4)):
        y, _ = run_gpu(torch_cuda, F, prog, x, variant=v)
        assert ndiff(y[:, 0, 0], want) == 0 and ndiff(y[:, 1, 0], want) == 0
    T4 = (x.shape[0] // 4) * 4
    xs = torch_cuda.from_numpy(np.ascontiguousarray(np.transpose(x[:T4], (1, 0, 2)))).cuda()
    ys, _ = prog.run_block_stream_major(xs)
    assert ndiff(ys[0, :, 0].cpu().numpy(), want[:T4]) == 0


@pytest.mark.parametrize("drive", ["dirac", "noise"])
@pytest.mark.parametrize("form", sorted(FORMS))
def test_reference_golden_vectors(torch_cuda, F, form, drive):
    x = bits(REF["inputs"][drive])
    prog = F.compile(F.from_sexpr(FORMS[form]()))
    y, _ = run_gpu(torch_cuda, F, prog, x[:, None, None])
    assert ndiff(y[:, 0, 0], bits(REF["outputs"][drive][form])) == 0


@pytest.mark.parametrize("drive", ["dirac", "noise"])
def test_reference_x_wire_two_outputs(torch_cuda, F, drive):
    x = bits(REF["inputs"][drive])
    y, _ = run_gpu(torch_cuda, F, F.compile(F.from_sexpr(G.cross_wire())), x[:, None, None])
    assert ndiff(y[:, 0, 0], bits(REF["outputs"][drive]["xwire0"])) == 0
    assert ndiff(y[:, 0, 1], bits(REF["outputs"][drive]["xwire1"])) == 0


# ---- every graph family vs the generic oracle, ragged sizes, all lane packings -------------------
GRAPHS = {
    "df1": G.df1, "df2": G.df2, "df1t": G.df1t, "df2t": G.df2t, "cascade6": lambda: G.df1_cascade(6),
    "integrator": G.integrator, "one_quad": G.one_quad, "one_quad_chain": G.one_quad_chain,
    "cross_wire": G.cross_wire, "par4": G.par4_sum, "par4_fanout": G.par4_sum_fanout,
    "identity": lambda: G.IN(1), "unit_delay": lambda: G.DEL(1, 1),
    "wire_around": lambda: ("seq", G.IN(1), G.IN(2)),
    "nested_fb": lambda: G.fb(G.seq(G.DEL(1, 1), G.fb(G.add(G.DEL(1, 1), G.IN(2))))),
    "long_delay_lds": lambda: G.seq(G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 40))),
                                    G.fb(G.add(G.mul(G.lit(0.7), G.DEL(1, 23)), G.IN(2)))),
    "depth8_regs": lambda: G.add(G.mul(G.lit(0.5), G.DEL(1, 8)), G.sub(G.DEL(1, 3), G.IN(1))),
    "div_neg": lambda: ("div", ("neg", G.IN(1)), G.add(G.lit(2.5), G.mul(G.DEL(1, 1), G.DEL(1, 1)))),
    "one_pole_double_literal": G.one_pole_readme,                      # flowz/README.md:52
    "mixed_precision_biquad": G.mixed_precision_biquad,
    "double_div": lambda: ("div", G.add(G.IN(1), G.lit64(1.5)), G.add(G.lit64(3.0), G.mul(G.DEL(1, 1), G.DEL(1, 1)))),
}


@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_graphs_vs_oracle_ragged(torch_cuda, F, name, P):
    g = GRAPHS[name]()
    prog = F.compile(F.from_sexpr(g))
    ns, T = 132, 101                      # 132 = 2 waves + 4 lanes: ragged last wave; T % unroll != 0
    x = O.synth_input(SEED + 1, np.arange(ns), T, n_wires=prog.n_in)    # n_in == 0: frames of width 0
    want = O.compile(g, ns).run(x)
    got, _ = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(P, 8))
    assert ndiff(got, want) == 0


@pytest.mark.parametrize("name", sorted(G.canonical_shape_bodies()))
def test_feedback_expressions_of_the_canonical_shape_tests(torch_cuda, F, name):
    """~X for every X of test/tests.cpp:26-60 (there: the TYPE un2bin gives the expression; here: what it computes -- no transform
    of that kind exists in this library, see tests/graphs.py:canonical_shape_bodies): time-major with two streams per lane and
    stream-major frames against the oracle."""
    g = G.fb(G.canonical_shape_bodies()[name])
    prog = F.compile(F.from_sexpr(g))
    ns, T = 70, 52
    assert (prog.n_in, prog.n_out) == (O.input_arity(g), O.output_arity(g))
    x = O.synth_input(SEED + 7, np.arange(ns), T, n_wires=prog.n_in)
    want = O.compile(g, ns).run(x)
    got, _ = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(2, 4))
    assert ndiff(got, want) == 0
    if prog.n_in:                                     # (the autonomous ones have no [stream][t] buffer to come in through)
        xs = torch_cuda.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        ys, _ = prog.run_block_stream_major(xs)
        assert ndiff(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0


@pytest.mark.parametrize("U", [1, 3, 4, 16, 32])
def test_unroll_variants_agree_with_oracle(torch_cuda, F, U):
    g = G.df1_cascade(3)
    prog = F.compile(F.from_sexpr(g))
    ns, T = 200, 77
    x = O.synth_input(5, np.arange(ns), T)
    got, _ = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(2, U))
    assert ndiff(got, C.df1_cascade([G.STABLE] * 3, x)) == 0


def test_single_stream_single_sample_edges(torch_cuda, F):
    prog = F.compile(F.from_sexpr(G.df1()))
    x = O.synth_input(9, [0], 1024)
    got, _ = run_gpu(torch_cuda, F, prog, x)                      # BASELINE config 1: 1 stream x 1024
    assert ndiff(got, C.df1_cascade([(G.B0, G.B1, G.B2, G.A1, G.A2)], x)) == 0
    got1, _ = run_gpu(torch_cuda, F, prog, x[:1])                 # one sample
    assert ndiff(got1, got[:1]) == 0


def test_block_chaining_and_state_layout(torch_cuda, F):
    """Blocks chain through the state buffer; its rows follow the documented layout."""
    from ir_interp import run_ir
    g = G.df1_cascade(6)
    prog = F.compile(F.from_sexpr(g))
    ns = 192
    x = O.synth_input(3, np.arange(ns), 300)
    whole, st_w = run_gpu(torch_cuda, F, prog, x)
    a, st = run_gpu(torch_cuda, F, prog, x[:123])
    b, st = run_gpu(torch_cuda, F, prog, x[123:], state=st)
    assert ndiff(np.concatenate([a, b]), whole) == 0
    assert ndiff(st.cpu().numpy(), st_w.cpu().numpy()) == 0
    _, st_ref = run_ir(prog, x)
    assert ndiff(st_w.cpu().numpy(), st_ref) == 0
    # LDS ring lines chain too, also when the split is not aligned with the ring size
    g2 = GRAPHS["long_delay_lds"]()
    p2 = F.compile(F.from_sexpr(g2))
    x2 = O.synth_input(4, np.arange(70), 150, n_wires=1)
    w2, _ = run_gpu(torch_cuda, F, p2, x2)
    a2, s2 = run_gpu(torch_cuda, F, p2, x2[:37])
    b2, s2 = run_gpu(torch_cuda, F, p2, x2[37:], state=s2)
    assert ndiff(np.concatenate([a2, b2]), w2) == 0 and ndiff(w2, O.compile(g2, 70).run(x2)) == 0


def test_osc_chain_per_stream_coefficients(torch_cuda, F):
    ns, T = 1000, 512
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    P = W.osc_chain_params(SEED + 1, np.arange(ns))
    x = np.zeros((T, ns, 1), np.float32)
    x[0] = 1.0
    for lanes in (1, 2, 4):
        got, _ = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(lanes, 8))
        assert ndiff(got, C.osc_chain(P, x)) == 0


def test_double_literal_graphs_vs_compiled_c(torch_cuda, F):
    """float64 sub-expressions (C++ double literals) at a few thousand streams, vs compiled C."""
    ns, T = 3000, 300
    x = O.synth_input(SEED + 9, np.arange(ns), T)
    p1 = F.compile(F.from_sexpr(G.one_pole_readme(0.9)))
    p2 = F.compile(F.from_sexpr(G.mixed_precision_biquad()))
    assert p1.n_const64 == 1 and p2.n_const64 == 3 and p1.stage_packable == 0
    for P in (1, 2, 4):
        got, _ = run_gpu(torch_cuda, F, p1, x, variant=F.make_variant(P, 8))
        assert ndiff(got, C.one_pole_readme(0.9, x)) == 0
        got, _ = run_gpu(torch_cuda, F, p2, x, variant=F.make_variant(P, 8))
        assert ndiff(got, C.mixed_precision_biquad(x)) == 0


def ndiff64(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return int((a.view(np.uint64) != b.view(np.uint64)).sum())


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_float64_output_frames_keep_double_results(torch_cuda, F, P):
    """FZ_VF_OUT_F64: the double result of a graph with double literals leaves un-narrowed (the
    tuple<double> of tests.cpp:222-231); float wires widen exactly.  vs compiled C and the oracle,
    time-major and tiled, chained blocks."""
    torch = torch_cuda
    ns, T = 2048, 150
    x = O.synth_input(SEED + 31, np.arange(ns), T)
    xd = torch.from_numpy(x).cuda()
    v = F.make_variant(P, 8) if P else None
    p1 = F.compile(F.from_sexpr(G.one_pole_readme(0.9)))
    p2 = F.compile(F.from_sexpr(G.mixed_precision_biquad()))
    for prog, want in ((p1, C.one_pole_readme(0.9, x, out_f64=True)), (p2, C.mixed_precision_biquad(x, out_f64=True))):
        y, _ = prog.run_block(xd, variant=v, out_f64=True)
        assert y.dtype == torch.float64 and ndiff64(y.cpu().numpy(), want) == 0
        assert (want != want.astype(np.float32)).any()                       # the low bits are really there
        ya, st = prog.run_block(xd[:70], variant=v, out_f64=True)            # chained blocks
        yb, _ = prog.run_block(xd[70:], state=st, variant=v, out_f64=True)
        assert ndiff64(torch.cat([ya, yb]).cpu().numpy(), want) == 0
        yt, _ = prog.run_block(F.to_tiled(xd, 256), variant=v, out_f64=True)  # tiled layout
        assert ndiff64(F.from_tiled(yt).cpu().numpy(), want) == 0
    # mixed output types: (float wire, double wire) -> both widened into one float64 frame
    g = G.seq(G.chan(G.IN(1), G.mul(G.lit64(0.3), G.IN(1))), G.par(G.add(G.IN(1), G.DEL(1, 1)), G.IN(1)))
    prog = F.compile(F.from_sexpr(g))
    assert prog.output_dtypes() == ["f32", "f64"]
    y, _ = prog.run_block(xd, variant=v, out_f64=True)
    want = O.compile(g, ns, out_f64=True).run(x)
    assert ndiff64(y.cpu().numpy(), want) == 0
    y32, _ = prog.run_block(xd, variant=v)
    assert ndiff(y32.cpu().numpy(), want.astype(np.float32)) == 0          # narrowing once == float32 frames


def test_float64_output_frames_of_float_graphs_and_stage_packing(torch_cuda, F):
    """A float-only graph widens exactly, also through the stage-packed kernel."""
    torch = torch_cuda
    ns, T = 512, 257
    x = O.synth_input(SEED + 32, np.arange(ns), T)
    xd = torch.from_numpy(x).cuda()
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    want = C.df1_cascade([G.STABLE] * 6, x).astype(np.float64)
    for v in (None, F.make_variant(1, 8, 256, F.C.FZ_VF_STAGE_PACK), F.make_variant(2, 4), F.make_variant(1, 8, 256, F.C.FZ_VF_NO_STAGE_PACK)):
        y, _ = prog.run_block(xd, variant=v, out_f64=True)
        assert ndiff64(y.cpu().numpy(), want) == 0


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_complex_wires_vs_std_complex(torch_cuda, F, P):
    """std::complex<float> wires (tests.cpp:206-207): every supported <complex> operator, next to a real
    feedback wire; frames hold (re, im, integrator).  vs the std::complex<float> spelling compiled by g++."""
    torch = torch_cuda
    ns, T = 4096, 120
    x = O.synth_input(SEED + 41, np.arange(ns), T)
    want = C.complex_mix(x, std=True)
    prog = F.compile(F.from_sexpr(G.complex_mix()))
    v = F.make_variant(P, 8) if P else None
    got, _ = run_gpu(torch, F, prog, x, variant=v)
    assert got.shape == (T, ns, 3) and ndiff(got, want) == 0
    y64, _ = prog.run_block(torch.from_numpy(x).cuda(), variant=v, out_f64=True)
    assert ndiff64(y64.cpu().numpy(), want.astype(np.float64)) == 0
    yt, _ = prog.run_block(F.to_tiled(torch.from_numpy(x).cuda(), 1024), variant=v)
    assert ndiff(F.from_tiled(yt).cpu().numpy(), want) == 0


def test_denormals_and_specials_are_kept(torch_cuda, F):
    """No flush-to-zero, NaN/Inf propagate like the CPU."""
    g = G.df1()
    prog = F.compile(F.from_sexpr(g))
    x = np.zeros((64, 8, 1), np.float32)
    x[0, 0] = 1e-38
    x[0, 1] = np.float32(1.5e-45)
    x[0, 2] = np.inf
    x[3, 3] = np.nan
    x[0, 4] = -0.0
    x[0, 5] = 3e38
    want = O.compile(g, 8).run(x)
    got, _ = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(2, 8))
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert ndiff(np.where(nan, 0, got), np.where(nan, 0, want)) == 0
    assert (np.abs(want[np.isfinite(want)]) < 1.2e-38).any(), "test must exercise denormals"


def test_set_const_changes_uniform_coefficient_between_blocks(torch_cuda, F):
    prog = F.compile(F.from_sexpr(G.df1()))
    x = O.synth_input(2, np.arange(64), 50)
    slot = [i for i, v in enumerate(prog.consts()) if np.float32(v) == G.B0][0]
    prog.set_const(slot, 0.5)
    got, _ = run_gpu(torch_cuda, F, prog, x)
    assert ndiff(got, C.df1_cascade([(0.5, G.B1, G.B2, G.A1, G.A2)], x)) == 0


def test_bad_arguments_fail_loudly(torch_cuda, F):
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1()))
    x = torch.zeros((8, 6, 1), device="cuda")
    with pytest.raises(F.FlowzError):
        prog.run_block(x, variant=F.make_variant(4, 8))          # 6 streams not a multiple of 4
    with pytest.raises(F.FlowzError):
        prog.run_block_ptr(x.data_ptr() + 4, x.data_ptr(), x.data_ptr(), None, 6, 8)   # misaligned
    with pytest.raises(F.NoDeviceError):
        prog.run_block(torch.zeros((8, 6, 1)))                   # host tensor: no CPU path


# ---- BASELINE sizes: sampled streams vs the compiled oracle + size-independent properties -------------
def _sample_ids(ns, k, seed):
    rng = np.random.default_rng(seed)
    ids = np.unique(np.concatenate([[0, 1, 63, 64, ns - 1], rng.integers(0, ns, k)]))
    return ids


def test_config2_cascade6_65536x4096_full_block(torch_cuda, F):
    """BASELINE config 2.  1024+ random streams, full length, bitwise vs the compiled oracle;
    all lane packings bit-identical on the WHOLE output; split blocks == one block."""
    torch = torch_cuda
    ns, T = 65536, 4096
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y1, st1 = prog.run_block(x, variant=F.make_variant(1, 8))
    ids = _sample_ids(ns, 1024, 1)
    xh = O.synth_input(SEED, ids, T)
    idt = torch.from_numpy(ids).cuda()
    assert ndiff(x[:, idt].cpu().numpy(), xh) == 0                # device generator == host generator
    want = C.df1_cascade([G.STABLE] * 6, xh)
    assert ndiff(y1[:, idt].cpu().numpy(), want) == 0
    assert np.isfinite(want).all()
    for P, U in ((2, 8), (4, 4), (2, 16)):
        y2, st2 = prog.run_block(x, variant=F.make_variant(P, U))
        assert torch.equal(y1.view(torch.int32), y2.view(torch.int32))
        assert torch.equal(st1.view(torch.int32), st2.view(torch.int32))
    # two half blocks chained through the state buffer
    ya, sta = prog.run_block(x[:2048])
    yb, stb = prog.run_block(x[2048:], state=sta)
    assert torch.equal(y1[:2048].view(torch.int32), ya.view(torch.int32))
    assert torch.equal(y1[2048:].view(torch.int32), yb.view(torch.int32))
    assert torch.equal(st1.view(torch.int32), stb.view(torch.int32))


def test_config3_par4_sum_1M_streams(torch_cuda, F):
    """BASELINE config 3 at full size: 4 parallel biquads summed, 1 M streams x 4096 samples (4 input wires per
    frame: 64 GiB of input frames, 16 GiB of output)."""
    torch = torch_cuda
    ns, T = 1 << 20, 4096
    prog = F.compile(F.from_sexpr(G.par4_sum()))
    x = torch.empty((T, ns, 4), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, _ = prog.run_block(x)
    ids = _sample_ids(ns, 1024, 2)
    xh = O.synth_input(SEED, ids, T, n_wires=4)
    idt = torch.from_numpy(ids).cuda()
    assert ndiff(x[:, idt].cpu().numpy(), xh) == 0
    assert ndiff(y[:, idt].cpu().numpy(), C.par4_sum(G.PAR4_SETS, xh)) == 0
    y1, _ = prog.run_block(x, variant=F.make_variant(1, 4))
    assert torch.equal(y.view(torch.int32), y1.view(torch.int32))
    del x, y, y1
    # fan-out variant: one input wire
    progf = F.compile(F.from_sexpr(G.par4_sum_fanout()))
    x1 = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x1, SEED)
    yf, _ = progf.run_block(x1)
    xh1 = O.synth_input(SEED, ids, T)
    assert ndiff(yf[:, idt].cpu().numpy(), C.par4_sum(G.PAR4_SETS, xh1, fanout=True)) == 0


def test_config4_osc_chain_1M_streams(torch_cuda, F):
    """BASELINE config 4 at full size: resonator oscillator -> 6 biquads, per-stream coefficients, 1 M streams x 4096."""
    torch = torch_cuda
    ns, T = 1 << 20, 4096
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    ids = _sample_ids(ns, 1024, 3)
    P = W.osc_chain_params(SEED + 1, np.arange(ns))
    pd = torch.from_numpy(P).cuda()
    x = torch.zeros((T, ns, 1), dtype=torch.float32, device="cuda")
    x[0] = 1.0
    y, _ = prog.run_block(x, params=pd)
    xh = np.zeros((T, len(ids), 1), np.float32)
    xh[0] = 1.0
    idt = torch.from_numpy(ids).cuda()
    want = C.osc_chain(np.ascontiguousarray(P[:, ids]), xh)
    assert ndiff(y[:, idt].cpu().numpy(), want) == 0
    assert np.isfinite(want).all() and np.abs(want[-1]).max() > 0      # still oscillating at the end


def test_config5_shard_equals_global_stream_ids(torch_cuda, F):
    """BASELINE config 5 shards streams across GPUs: a shard computed alone (stream0 offset in
    the generator) is bit-identical to the same streams inside one big batch."""
    torch = torch_cuda
    ns, T, shard = 8192, 256, 2048
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, _ = prog.run_block(x)
    for r in range(ns // shard):
        xs = torch.empty((T, shard, 1), dtype=torch.float32, device="cuda")
        F.synth_fill(xs, SEED, stream0=r * shard)
        ys, _ = prog.run_block(xs)
        assert torch.equal(ys.view(torch.int32), y[:, r * shard:(r + 1) * shard].contiguous().view(torch.int32))


# ---- typed programs: ResultType through inputs, state and outputs ---------------------------------------
def _typed_gpu(torch, F, prog, frames_host, variant=None, state=None, tile=0, stream_major=False):
    x = torch.from_numpy(np.ascontiguousarray(frames_host)).cuda()
    if stream_major:
        y, st = prog.run_block_stream_major(x.permute(1, 0, 2).contiguous(), state=state, variant=variant)
        return y.permute(1, 0, 2).contiguous().cpu().numpy(), st
    if tile:
        y, st = prog.run_block(F.to_tiled(x, tile), state=state, variant=variant)
        return F.from_tiled(y).contiguous().cpu().numpy(), st
    y, st = prog.run_block(x, state=state, variant=variant)
    return y.cpu().numpy(), st


@pytest.mark.parametrize("ns", [256, 768, 2048, 5120])
def test_lane_groups_vs_oracle(torch_cuda, F, ns):
    """Lane groups (round 5, internal flags FZ_VF_LANE_PAIRS / _SINGLES; letters L / S in the kernel name): whole waves of 64 P adjacent
    streams whose lanes take their P streams in groups 64 x group size apart, so that no access of a lane is wider than 16 bytes.  Typed
    frames of 8 bytes per stream with four streams per lane (pairs), 4-wire frames with two and four streams per lane (singles), a
    2-in / 2-out float graph (pairs): free-running and in lockstep, chained blocks, state bit-identical to the adjacent-streams kernels."""
    torch = torch_cuda
    T = 101
    L, GS, P3 = F.C.FZ_VF_LOCKSTEP, F.C.FZ_VF_GRID_SYNC, F.C.FZ_VF_PREFETCH3
    x1 = O.synth_input(SEED + 71, np.arange(ns), T)
    # typed graphs: (graph, oracle output as packed frames)
    typed = [(G.complex_one_pole(), C.complex_one_pole(x1, std=True)), (G.complex_div_mix(), C.complex_div_mix(x1, std=True))]
    gd = W.df1_double() if hasattr(W, "df1_double") else None
    for g, want in typed:
        prog = F.compile(F.from_sexpr(g), typed=True)
        ref, st_ref = _typed_gpu(torch, F, prog, x1, F.make_variant(1, 8))
        assert ndiff(ref, want) == 0
        for v in ((4, 8, 256, 0), (4, 4, 64, 0), (4, 1, 256, L | P3), (4, 2, 128, L | GS), (4, 16, 0, 0)):
            if ns % 256:
                continue
            vv = F.make_variant(*v)
            assert prog.kernel_name(vv, ns, T).endswith("L"), prog.kernel_name(vv, ns, T)
            got, st = _typed_gpu(torch, F, prog, x1, vv)
            assert ndiff(got, want) == 0 and torch.equal(st.view(torch.int32), st_ref.view(torch.int32)), (ns, v)
            if ns % 1024 == 0 and v[2] in (0, 256):                                                 # stream tiles of 1024: one 256-lane workgroup of pairs per tile
                assert prog.kernel_name(vv, ns, T, 1024).endswith("L")
                got_t, st_t = _typed_gpu(torch, F, prog, x1, vv, tile=1024)
                assert ndiff(got_t, want) == 0 and torch.equal(st_t.view(torch.int32), st_ref.view(torch.int32)), (ns, v, "tiled")
            a, st1 = _typed_gpu(torch, F, prog, x1[:40], vv)
            b, st2 = _typed_gpu(torch, F, prog, x1[40:], F.make_variant(2, 8), state=st1)           # ... continued by an adjacent-streams kernel
            assert ndiff(np.concatenate([a, b]), want) == 0 and torch.equal(st2.view(torch.int32), st_ref.view(torch.int32))
    # a double biquad (double state rows: fz_ld_row64 / fz_st_row64 in groups)
    from zignal_amd import workloads as ZW
    g = ZW.df1_double()
    prog = F.compile(F.from_sexpr(g), typed=True)
    want = O.run_typed(O.compile(g, ns, typed=True), [x1[:, :, 0]])
    ref, st_ref = _typed_gpu(torch, F, prog, x1, F.make_variant(2, 8))
    for a_, b_ in zip(F.unpack_typed(ref, prog.output_dtypes()), want):
        assert np.array_equal(a_, b_)
    for v in ((4, 8, 256, 0), (4, 1, 256, L | P3)):
        vv = F.make_variant(*v)
        assert prog.kernel_name(vv, ns, T).endswith("L")
        got, st = _typed_gpu(torch, F, prog, x1, vv)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and torch.equal(st.view(torch.int32), st_ref.view(torch.int32)), (ns, v)
    # 4-wire frames (the 4-parallel sum): singles with two and four streams per lane
    g = G.par4_sum()
    prog = F.compile(F.from_sexpr(g))
    x4 = O.synth_input(SEED + 72, np.arange(ns), T, n_wires=4)
    want = O.compile(g, ns).run(x4)
    ref, st_ref = run_gpu(torch, F, prog, x4, variant=F.make_variant(1, 8))
    assert ndiff(ref, want) == 0
    for v in ((2, 8, 128, 0), (2, 1, 128, L | P3), (2, 2, 128, L | GS), (4, 4, 64, 0), (2, 16, 0, 0)):
        if ns % (64 * v[0]):
            continue
        vv = F.make_variant(*v)
        assert prog.kernel_name(vv, ns, T).endswith("S"), prog.kernel_name(vv, ns, T)
        got, st = run_gpu(torch, F, prog, x4, variant=vv)
        assert ndiff(got, want) == 0 and torch.equal(st, st_ref), (ns, v)
    # 2 wires in, 2 wires out (two biquads side by side): pairs with four streams per lane
    g = G.par(G.df1(*G.STABLE), G.df1(*G.PAR4_SETS[1]))
    prog = F.compile(F.from_sexpr(g))
    x2 = O.synth_input(SEED + 73, np.arange(ns), T, n_wires=2)
    want = O.compile(g, ns).run(x2)
    for v in ((4, 8, 256, 0), (4, 2, 64, L)):
        if ns % 256:
            continue
        vv = F.make_variant(*v)
        assert prog.kernel_name(vv, ns, T).endswith("L")
        got, _ = run_gpu(torch, F, prog, x2, variant=vv)
        assert ndiff(got, want) == 0, (ns, v)
    # graphs WITHOUT input wires (generators): a typed complex rotor (pairs) and a 4-output float generator (singles) -- the lane's frame slice
    # on the input side is empty (round 5: the first cut of the lane groups did not compile for these)
    gens = [(G.fb(G.add(G.mul(G.litc(0.6, 0.7), G.DEL(1, 1)), G.litc(0.001, 0.0))), True, (4, 8, 256, 0), "L"),
            (G.chan(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.lit(0.1))), G.lit(1.0), G.lit(2.0), G.lit(3.0)), False, (2, 8, 128, 0), "S")]
    for g, is_typed, v, letter in gens:
        if ns % (64 * v[0]):
            continue
        prog = F.compile(F.from_sexpr(g), typed=is_typed)
        assert prog.n_in == 0
        x0 = torch.zeros((T, ns, 0), dtype=torch.float32, device="cuda")
        ref, st_ref = prog.run_block(x0, variant=F.make_variant(1, 8))
        vv = F.make_variant(*v)
        assert prog.kernel_name(vv, ns, T).endswith(letter)
        got, st = prog.run_block(x0, variant=vv)
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)) and torch.equal(st.view(torch.int32), st_ref.view(torch.int32)), (ns, v)
        assert float(got.abs().max()) > 0
        if not is_typed:
            want = O.compile(g, ns).run(np.zeros((T, ns, 0), np.float32))
            assert ndiff(got.cpu().numpy(), want) == 0
    # float64 output frames of a float graph with a double literal (FZ_VF_OUT_F64): 8 bytes per stream out -> pairs with four streams per lane
    if ns % 256 == 0:
        g = G.mixed_precision_biquad()
        prog = F.compile(F.from_sexpr(g))
        xd = torch.from_numpy(x1).cuda()
        ref64, st_ref = prog.run_block(xd, variant=F.make_variant(1, 8), out_f64=True)
        want64 = O.compile(g, ns, out_f64=True).run(x1)
        assert np.array_equal(ref64.cpu().numpy().view(np.uint64), np.ascontiguousarray(want64, np.float64).view(np.uint64))
        for v in ((4, 8, 256, 0), (4, 2, 64, L)):
            vv = F.make_variant(v[0], v[1], v[2], v[3] | F.C.FZ_VF_OUT_F64)
            assert prog.kernel_name(vv, ns, T).endswith("L"), prog.kernel_name(vv, ns, T)
            got64, st = prog.run_block(xd, variant=F.make_variant(*v), out_f64=True)
            assert torch.equal(got64.view(torch.int64), ref64.view(torch.int64)) and torch.equal(st, st_ref), (ns, v)
    # the developer switch of the comparison kernels leaves the names bare
    assert not F.compile(F.from_sexpr(G.df1_cascade(2))).kernel_name(F.make_variant(4, 8, 256), ns, T).endswith(("L", "S"))


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_typed_programs_complex_state_division_double_state(torch_cuda, F, P):
    """fz_compile_typed on the GPU vs std::complex<float> compiled by g++ / compiled C with double state: the complex
    one-pole ~(c*_1[_1] + _2) (complex delay line), z/w and s/w (__divsc3), the double accumulator of tests.cpp:223;
    every lane packing, ragged stream counts, chained blocks, tiled and stream-major frames."""
    torch = torch_cuda
    ns, T = 776, 93
    v = F.make_variant(P, 8) if P else None
    x = O.synth_input(SEED + 61, np.arange(ns), T)
    for g, want in ((G.complex_one_pole(), C.complex_one_pole(x, std=True)), (G.complex_div_mix(), C.complex_div_mix(x, std=True))):
        prog = F.compile(F.from_sexpr(g), typed=True)
        got, st = _typed_gpu(torch, F, prog, x, v)
        assert ndiff(got, want) == 0
        a, st1 = _typed_gpu(torch, F, prog, x[:40], v)                         # two blocks chained through the typed state
        b, st2 = _typed_gpu(torch, F, prog, x[40:], v, state=st1)
        assert ndiff(np.concatenate([a, b]), want) == 0 and torch.equal(st2, st)
    prog = F.compile(F.from_sexpr(G.double_accumulator()), typed=True)
    want = C.double_accumulator(x)[:, :, 0]
    got, st = _typed_gpu(torch, F, prog, x, v)
    assert np.array_equal(F.unpack_typed(got, ["f64"])[0], want)
    assert np.array_equal(st.cpu().numpy().reshape(-1)[:2 * ns].view(np.float64), want[-1])      # state row 0 = ns doubles
    a, st1 = _typed_gpu(torch, F, prog, x[:17], v)
    b, _ = _typed_gpu(torch, F, prog, x[17:], v, state=st1)
    assert np.array_equal(F.unpack_typed(np.concatenate([a, b]), ["f64"])[0], want)
    if P in (0, 1):
        for kw in ({"tile": 0, "stream_major": True},):
            got, _ = _typed_gpu(torch, F, prog, x[:92], None, **kw)             # (stream-major rows: multiples of 4 floats)
            assert np.array_equal(F.unpack_typed(got, ["f64"])[0], want[:92])
    with pytest.raises(F.FlowzError):
        prog.run_block(torch.zeros((4, 8, 1), device="cuda"), out_f64=True)    # FZ_VF_OUT_F64 does not apply to typed programs
    # a double delay line of 12 samples: an LDS ring of (low word, high word) pairs, next to a float ring of 20
    g = G.chan(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 12)), G.mul(G.lit64(0.25), G.IN(2)))), G.add(G.IN(1), G.DEL(1, 20)))
    prog = F.compile(F.from_sexpr(g), typed=True)
    assert prog.line_dtypes() == ["f64", "f32"]
    want = O.run_typed(O.compile(g, ns, typed=True), [x[:, :, 0]])
    got, st = _typed_gpu(torch, F, prog, x, v)
    for a_, b_ in zip(F.unpack_typed(got, prog.output_dtypes()), want):
        assert a_.dtype == b_.dtype and np.array_equal(a_, b_)
    a, st1 = _typed_gpu(torch, F, prog, x[:31], v)
    b, st2 = _typed_gpu(torch, F, prog, x[31:], v, state=st1)
    assert np.array_equal(np.concatenate([a, b]).view(np.uint32), got.view(np.uint32)) and torch.equal(st2.view(torch.int32), st.view(torch.int32))


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_typed_programs_double_and_complex_input_frames(torch_cuda, F, P):
    """double and std::complex<float> INPUT wires (the reference's callable is a template over its argument types,
    flowz.hpp:1225-1229): frames carry them in two float slots; mixed float / double lines in the state; vs the typed oracle."""
    torch = torch_cuda
    ns, T = 512, 41
    g = G.chan(G.chan(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 2)), G.IN(2))),                        # double in -> double loop, depth 2
                      G.add(G.mul(G.IN(2), G.litc(0.25, -0.5)), G.DEL(2, 3))),                      # complex in, its own delayed value
               G.fb(G.add(G.mul(G.lit(0.75), G.DEL(1, 1)), G.mul(G.IN(4), G.IN(4)))))               # float in -> float loop
    dts = ["f64", "cf32", "f32"]
    prog = F.compile(F.from_sexpr(g), in_dtypes=dts)
    assert prog.output_dtypes() == dts and prog.line_dtypes().count("f64") == 1 and (prog.n_in, prog.n_out) == (5, 5)
    rng = np.random.default_rng(7)
    w = [rng.standard_normal((T, ns)), (rng.standard_normal((T, ns)) + 1j * rng.standard_normal((T, ns))).astype(np.complex64),
         rng.standard_normal((T, ns)).astype(np.float32)]
    want = O.run_typed(O.compile(g, ns, typed=True, in_dtypes=dts), w)
    fr = F.pack_typed(w, dts)
    v = F.make_variant(P, 4) if P else None
    got, st = _typed_gpu(torch, F, prog, fr, v)
    for a, b in zip(F.unpack_typed(got, dts), want):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    got_t, st_t = _typed_gpu(torch, F, prog, fr, F.make_variant(P, 4, 64) if P else None, tile=256)        # (a tile holds whole workgroups)
    # (state rows of double lines hold double words: compare bit patterns, a float32 view may read as NaN)
    assert np.array_equal(got_t.view(np.uint32), got.view(np.uint32)) and torch.equal(st.view(torch.int32), st_t.view(torch.int32))
    a, st1 = _typed_gpu(torch, F, prog, fr[:19], v)
    b, st2 = _typed_gpu(torch, F, prog, fr[19:], v, state=st1)
    assert np.array_equal(np.concatenate([a, b]).view(np.uint32), got.view(np.uint32)) and torch.equal(st2.view(torch.int32), st.view(torch.int32))


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_typed_programs_complex_double_state_and_smith_division(torch_cuda, F, P):
    """std::complex<double> wires on the GPU: complex<double> state (two double lines), z / w and s / w (__divdc3: Smith's
    method, both sides evaluated and selected per stream) vs std::complex<double> compiled by g++; complex<double> input
    frames (four slots) vs the typed oracle; time-major, stream-tiled, stream-major, chained blocks."""
    from test_oracle_c import _cdouble_input
    torch = torch_cuda
    ns, T = 1024, 64
    x = _cdouble_input(T, ns)
    want = C.cdouble_resonator(x, std=True)
    g = G.cdouble_resonator()
    prog = F.compile(F.from_sexpr(g), in_dtypes=["f64"])
    fr = F.pack_typed([x], ["f64"])
    v = F.make_variant(P, 4) if P else None
    got, st = _typed_gpu(torch, F, prog, fr, v)
    assert np.array_equal(F.unpack_typed(got, ["cf64"])[0].view(np.int64), want.view(np.int64))
    a, st1 = _typed_gpu(torch, F, prog, fr[:23], v)
    b, st2 = _typed_gpu(torch, F, prog, fr[23:], v, state=st1)
    assert np.array_equal(np.concatenate([a, b]).view(np.uint32), got.view(np.uint32)) and torch.equal(st2.view(torch.int32), st.view(torch.int32))
    got_t, st_t = _typed_gpu(torch, F, prog, fr, F.make_variant(P, 4, 64) if P else None, tile=256)
    assert np.array_equal(got_t.view(np.uint32), got.view(np.uint32)) and torch.equal(st.view(torch.int32), st_t.view(torch.int32))
    if P in (0, 1):
        got_s, st_s = _typed_gpu(torch, F, prog, fr, None, stream_major=True)
        assert np.array_equal(got_s.view(np.uint32), got.view(np.uint32)) and torch.equal(st.view(torch.int32), st_s.view(torch.int32))
    # complex<double> input wires with the other three types, a complex<double> wire through a depth-3 line
    g = G.chan(G.chan(("div", G.IN(1), G.add(G.IN(2), G.DEL(1, 3))), G.mul(G.IN(3), G.IN(4))), G.sub(G.IN(2), G.IN(1)))
    dts = ["cf64", "f64", "cf32", "f32"]
    prog = F.compile(F.from_sexpr(g), in_dtypes=dts)
    rng = np.random.default_rng(13)
    cplx = lambda dt: (rng.standard_normal((T, ns)) + 1j * rng.standard_normal((T, ns))).astype(dt)   # noqa: E731
    w = [cplx(np.complex128), rng.standard_normal((T, ns)), cplx(np.complex64), rng.standard_normal((T, ns)).astype(np.float32)]
    want = O.run_typed(O.compile(g, ns, typed=True, in_dtypes=dts), w)
    got, _ = _typed_gpu(torch, F, prog, F.pack_typed(w, dts), v)
    for a, b in zip(F.unpack_typed(got, prog.output_dtypes()), want):
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("case", KA["result_types"], ids=lambda c: "tests.cpp:" + c["lines"])
def test_typed_programs_tests_cpp_result_types_on_gpu(torch_cuda, F, case):
    """test_result_type_transform (tests.cpp:184-232) evaluated on the GPU under fz_compile_typed: the output frames carry
    exactly the types the reference asserts (tests.cpp:219 included) and the values of the typed oracle."""
    g = tup(case["graph"])
    prog = F.compile(F.from_sexpr(g), typed=True)
    assert prog.output_dtypes() == case.get("result_type", case["types"])
    ns, T = 70, 21
    x = np.ascontiguousarray(O.synth_input(SEED + 62, np.arange(ns), T, n_wires=max(prog.n_in, 1))[:, :, :prog.n_in])   # (a graph may have no input)
    want = O.run_typed(O.compile(g, ns, typed=True), [x[:, :, i] for i in range(prog.n_in)], T=T)
    got, _ = _typed_gpu(torch_cuda, F, prog, x)
    for a, b in zip(F.unpack_typed(got, prog.output_dtypes()), want):
        assert a.dtype == b.dtype and np.array_equal(a, b)


# ---- stream-tiled frames (fz_run_block_tiled) ---------------------------------------------------------
@pytest.mark.parametrize("P,tile", [(1, 256), (2, 512), (4, 1024), (0, 2048), (2, 1024)])
def test_tiled_layout_equals_time_major(torch_cuda, F, P, tile):
    torch = torch_cuda
    ns, T = 4096, 77
    for g in (G.df1_cascade(6), G.par4_sum(), G.cross_wire()):
        prog = F.compile(F.from_sexpr(g))
        x = torch.empty((T, ns, prog.n_in), dtype=torch.float32, device="cuda")
        F.synth_fill(x, SEED)
        y, st = prog.run_block(x, variant=F.make_variant(P, 8))
        xt = torch.empty((ns // tile, T, tile, prog.n_in), dtype=torch.float32, device="cuda")
        F.synth_fill(xt, SEED)                                         # tiled placement of the same values
        assert torch.equal(xt, F.to_tiled(x, tile))
        yt, stt = prog.run_block(xt, variant=F.make_variant(P, 8))
        assert torch.equal(F.from_tiled(yt).contiguous().view(torch.int32), y.view(torch.int32))
        assert torch.equal(stt.view(torch.int32), st.view(torch.int32))
    want = O.compile(G.cross_wire(), 8).run(O.synth_input(SEED, np.arange(8), T))
    assert ndiff(F.from_tiled(yt)[:, :8].cpu().numpy(), want) == 0


def test_tiled_layout_rejects_bad_tiles(torch_cuda, F):
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1()))
    x = torch.zeros((4, 8, 96, 1), device="cuda")                       # tile of 96 streams: not a multiple of 64
    with pytest.raises(F.FlowzError):
        prog.run_block(x)


# ---- stage packing (FZ_VF_STAGE_PACK): halves of a serial graph in one v_pk_*, B one sample behind A ----
STAGE_PACK = 8
NO_STAGE_PACK = 16
PACKABLE = {
    "cascade6": lambda: G.df1_cascade(6),
    "cascade8": lambda: G.df1_cascade(8),
    "cascade3_prefix_plus_2": lambda: G.df1_cascade(3),
    "cascade5_prefix_plus_4": lambda: G.df1_cascade(5),
    "cascade7_prefix_plus_6": lambda: G.df1_cascade(7, [G.STABLE, G.PAR4_SETS[0], G.PAR4_SETS[1], G.STABLE, G.PAR4_SETS[3], G.STABLE, G.PAR4_SETS[2]]),
    "integrator_then_cascade2": lambda: G.seq(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))), G.df1_cascade(2)),
    "cascade12_two_stages_per_segment": lambda: G.df1_cascade(12),
    "cascade2": lambda: G.df1_cascade(2),
    "one_quad_chain": G.one_quad_chain,
    "cascade4_distinct_coeffs": lambda: G.df1_cascade(4, [G.STABLE, G.PAR4_SETS[0], G.PAR4_SETS[1], G.PAR4_SETS[2]]),
    "df2_pair": lambda: G.seq(G.df2(*G.STABLE), G.df2(*G.PAR4_SETS[3])),
    # scalar SUFFIX behind the chain: output gain, a smoothing one-pole with its own state, a mix with the
    # chain's own delayed output (a delayed read shared with the last stage), prefix + suffix together
    "cascade6_output_gain": lambda: G.seq(G.df1_cascade(6), G.mul(G.lit(0.7), G.IN(1))),
    "cascade4_smoothing_one_pole": lambda: G.seq(G.df1_cascade(4), G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.mul(G.lit(0.5), G.IN(2))))),
    "cascade6_mix_with_delayed_output": lambda: G.seq(G.df1_cascade(6), G.add(G.mul(G.lit(0.6), G.IN(1)), G.mul(G.lit(0.3), G.DEL(1, 2)))),
    "cascade4_suffix_delay_beyond_the_chain": lambda: G.seq(G.df1_cascade(4), G.sub(G.IN(1), G.mul(G.lit(0.25), G.DEL(1, 5)))),
    "integrator_cascade4_gain": lambda: G.seq(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))), G.seq(G.df1_cascade(4), G.mul(G.IN(1), G.lit(1.5)))),
}


@pytest.mark.parametrize("T", [1, 2, 3, 5, 6, 7, 9, 17, 101])
@pytest.mark.parametrize("name", sorted(PACKABLE))
def test_stage_packed_kernel_vs_oracle(torch_cuda, F, name, T):
    g = PACKABLE[name]()
    prog = F.compile(F.from_sexpr(g))
    assert prog.stage_packable == 1
    ns = 133
    x = O.synth_input(SEED + 2, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    got, st = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, 8, 256, STAGE_PACK))
    assert ndiff(got, want) == 0
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0       # canonical state after the epilogue


@pytest.mark.parametrize("T", [1, 4, 7, 64])
def test_stage_packed_osc_chain_with_scalar_prefix_and_per_stream_coefficients(torch_cuda, F, T):
    """resonator (scalar prefix) -> 6 DF1 stages (6 packed segments), 31 per-stream coefficients."""
    ns = 200
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    assert prog.stage_packable == 1
    P = W.osc_chain_params(SEED + 3, np.arange(ns))
    x = np.zeros((T, ns, 1), np.float32)
    x[0] = 1.0
    got, st = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(1, 8, 256, STAGE_PACK))
    assert ndiff(got, C.osc_chain(P, x)) == 0
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0
    # next block continues from the canonical state with the other variant
    x2 = np.zeros((9, ns, 1), np.float32)
    a, _ = run_gpu(torch_cuda, F, prog, x2, params=P, variant=F.make_variant(2, 8), state=st)
    b, _ = run_gpu(torch_cuda, F, prog, x2, params=P, variant=F.make_variant(1, 8, 256, STAGE_PACK), state=st_ref)
    assert ndiff(a, b) == 0


def test_stage_packed_blocks_chain_with_any_variant(torch_cuda, F):
    g = G.df1_cascade(6)
    prog = F.compile(F.from_sexpr(g))
    ns = 256
    x = O.synth_input(8, np.arange(ns), 120)
    want = C.df1_cascade([G.STABLE] * 6, x)
    sk, plain, packed2 = F.make_variant(1, 16, 256, STAGE_PACK), F.make_variant(1, 8, 256, NO_STAGE_PACK), F.make_variant(2, 8)
    a, st = run_gpu(torch_cuda, F, prog, x[:33], variant=sk)
    b, st = run_gpu(torch_cuda, F, prog, x[33:34], variant=sk, state=st)       # a 1-sample block
    c, st = run_gpu(torch_cuda, F, prog, x[34:70], variant=plain, state=st)
    d, st = run_gpu(torch_cuda, F, prog, x[70:99], variant=sk, state=st)
    e, st = run_gpu(torch_cuda, F, prog, x[99:], variant=packed2, state=st)
    assert ndiff(np.concatenate([a, b, c, d, e]), want) == 0


def test_stage_pack_is_automatic_for_few_streams_and_rejected_when_impossible(torch_cuda, F):
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    src = prog.source(F.make_variant(1, 16, 256, STAGE_PACK))
    assert "#define FZ_SKEW 5" in src and "#define FZ_NSEG 6" in src and "step2" in src     # 6 stages = 6 segments (one atom each: only graphs of one or two packed pairs are cut further)
    x = torch.zeros((4, 64, 1), device="cuda")
    with pytest.raises(F.FlowzError):
        F.compile(F.from_sexpr(G.par4_sum_fanout())).run_block(x, variant=F.make_variant(1, 8, 256, STAGE_PACK))
    with pytest.raises(F.FlowzError):
        prog.run_block(x, variant=F.make_variant(2, 8, 256, STAGE_PACK))


# ---- wave split: two waves per 64 streams, each one half of the serial graph (fewer streams than lanes) ---------------
WAVE_SPLIT = 1024
SPLITTABLE = {
    "cascade4": lambda: G.df1_cascade(4),                                          # halves: one packed pair each
    "cascade6": lambda: G.df1_cascade(6),                                          # halves: a packed pair and a scalar stage
    "cascade8": lambda: G.df1_cascade(8),                                          # halves: two packed pairs, lag 3
    "cascade16": lambda: G.df1_cascade(16),                                        # halves: four packed pairs, lag 7 (hand-off distance 8)
    "cascade4_distinct_coeffs": lambda: G.df1_cascade(4, [G.STABLE, G.PAR4_SETS[0], G.PAR4_SETS[1], G.PAR4_SETS[2]]),
    "cascade6_distinct_coeffs": lambda: G.df1_cascade(6, [G.STABLE, G.PAR4_SETS[0], G.PAR4_SETS[1], G.PAR4_SETS[2], G.PAR4_SETS[3], G.STABLE]),
    "df2_x4": lambda: G.seq(G.seq(G.df2(*G.STABLE), G.df2(*G.PAR4_SETS[3])), G.seq(G.df2(*G.PAR4_SETS[1]), G.df2(*G.STABLE))),
    # a scalar prefix goes with the first part, a scalar suffix with the last one
    "cascade7_prefix_stage": lambda: G.df1_cascade(7),
    "cascade6_output_gain": lambda: G.seq(G.df1_cascade(6), G.mul(G.lit(0.7), G.IN(1))),
    "integrator_cascade4_gain": lambda: G.seq(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))), G.seq(G.df1_cascade(4), G.mul(G.IN(1), G.lit(1.5)))),
    "cascade4_smoothing_one_pole": lambda: G.seq(G.df1_cascade(4), G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.mul(G.lit(0.5), G.IN(2))))),
}


@pytest.mark.parametrize("T", [1, 2, 3, 5, 13, 16, 17, 40, 64, 101, 300])
@pytest.mark.parametrize("name", sorted(SPLITTABLE))
def test_wave_split_kernel_vs_oracle(torch_cuda, F, name, T):
    """FZ_VF_WAVES(2 / 3 / 4): the cut wires go through LDS, every wave runs a round behind the one before; every block length
    (all-masked blocks, blocks that end inside a round, many rounds), ragged stream counts, unroll 8 / 16 / 32, one or two
    tuples per workgroup -- outputs and the canonical state against the oracle / the plain kernel."""
    g = SPLITTABLE[name]()
    prog = F.compile(F.from_sexpr(g))
    ns = 133
    x = O.synth_input(SEED + 5, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert ndiff(ref, want) == 0
    tried = 0
    for W, io in ((2, 0), (3, 0), (4, 0), (1, 1), (2, 1), (3, 1), (1, 2), (2, 2), (3, 2)):   # parts = compute waves per 64 streams (whatever the graph divides into), + one / two I/O waves
        fl = F.C.FZ_VF_WAVES(W) | (F.C.FZ_VF_IO_WAVE if io else 0) | (F.C.FZ_VF_IO_WAVE2 if io == 2 else 0)
        try:
            prog.kernel_name(F.make_variant(1, 16, 0, fl), ns, T)
        except F.FlowzError:
            continue
        for U, B in ((8, 64), (16, 128), (32, 0), (16, 64)):
            got, st = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, U, B, fl))
            assert ndiff(got, want) == 0, (name, T, W, io, U, B)
            assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0, (name, T, W, io, U, B)
            tried += 1
    assert tried >= 8


@pytest.mark.parametrize("T", [1, 7, 64, 101, 300])
@pytest.mark.parametrize("name", sorted(PACKABLE))
def test_io_wave_kernel_vs_oracle(torch_cuda, F, name, T):
    """FZ_VF_IO_WAVE alone: every stage-packable graph (scalar prefix / suffix included) as one compute wave next to an I/O
    wave -- the input rows reach it through LDS two rounds after they were requested, the output rows leave the same way."""
    g = PACKABLE[name]()
    prog = F.compile(F.from_sexpr(g))
    ns = 197
    x = O.synth_input(SEED + 6, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    for U, B in ((16, 0), (8, 128), (32, 64)):
        got, st = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, U, B, F.C.FZ_VF_IO_WAVE))
        assert ndiff(got, want) == 0, (name, T, U, B)
        assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0, (name, T, U, B)


def test_io_wave_with_per_stream_coefficients(torch_cuda, F):
    """resonator (scalar prefix) -> 6 DF1 stages with 31 per-stream coefficients (config 4's graph) behind an I/O wave"""
    ns = 200
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    P = W.osc_chain_params(SEED + 3, np.arange(ns))
    x = np.zeros((333, ns, 1), np.float32)
    x[0] = 1.0
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    for io in (F.C.FZ_VF_IO_WAVE, F.C.FZ_VF_IO_WAVE | F.C.FZ_VF_IO_WAVE2):        # one I/O wave; a loader and a storer
        got, st = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(1, 16, 0, io))
        assert ndiff(got, C.osc_chain(P, x)) == 0
        assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0


def test_two_io_waves_are_the_default_between_32768_and_65536_streams(torch_cuda, F):
    """Round 6: from 32 769 to 65 536 streams (at most one wave per SIMD of work) a stage-packable graph of <= 64 operations runs as ONE compute wave per 64
    streams next to a loader and a storer wave (ahead of the lone wave in paired bursts on four boards: profiles/r06/config2_io_waves_default.txt).  The
    library's default by name on rows and tiles; a stream count that fills neither the last wave nor the last workgroup; two chained blocks and a window
    of longer buffers; the oscillator chain with its per-stream coefficients: sampled streams against the oracle, everything against the lone wave."""
    torch = torch_cuda
    IO2 = F.C.FZ_VF_IO_WAVE | F.C.FZ_VF_IO_WAVE2
    g = G.df1_cascade(6)
    prog = F.compile(F.from_sexpr(g))
    ns, T = 40000 + 37, 1100
    assert prog.kernel_name(None, ns, T) == "fz_block_kernel_p1u16b256w1io2f%dM" % IO2 and prog.kernel_name(None, 65536, 4096, 8192) == "fz_block_kernel_p1u16b256w1io2f%d" % IO2
    lone = F.make_variant(1, 16, 0, F.C.FZ_VF_STAGE_PACK)
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 93)
    y1, st1 = prog.run_block(x[:600].contiguous())
    y2, st2 = prog.run_block(x[600:].contiguous(), state=st1.clone())
    r1, sr1 = prog.run_block(x[:600].contiguous(), variant=lone)
    r2, sr2 = prog.run_block(x[600:].contiguous(), state=sr1.clone(), variant=lone)
    assert torch.equal(y1, r1) and torch.equal(y2, r2) and torch.equal(st1, sr1) and torch.equal(st2, sr2)
    ids = np.concatenate([np.arange(3), np.random.default_rng(3).integers(0, ns, 90), np.arange(ns - 40, ns)])
    want = C.df1_cascade([G.STABLE] * 6, O.synth_input(SEED + 93, ids, T))
    assert ndiff(torch.cat([y1, y2])[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), want) == 0
    # a window of the same buffers (rows 300 .. 899), state carried from a block over the rows before it
    out = torch.zeros_like(x)
    st = torch.zeros((prog.n_state, ns), device="cuda")
    prog.run_window(x, out, st, 0, 300)
    prog.run_window(x, out, st, 300, 600)
    assert torch.equal(out[:900], torch.cat([y1, y2])[:900])
    # tiles
    nt, tile = 65536, 8192
    xt = torch.empty((nt // tile, 700, tile, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(xt, SEED + 94)
    yt, stt = prog.run_block(xt)
    rt, srt = prog.run_block(xt, variant=lone)
    assert torch.equal(yt, rt) and torch.equal(stt, srt)
    # per-stream coefficients ride along (config 4's graph at 50 000 streams)
    po = F.compile(F.from_sexpr(G.osc_chain(6)))
    no = 50000
    assert po.kernel_name(None, no, 1024).startswith("fz_block_kernel_p1u16b256w1io2f")
    params = torch.from_numpy(W.osc_chain_params(SEED + 1, np.arange(no))).cuda()
    xo = torch.zeros((1024, no, 1), dtype=torch.float32, device="cuda")
    xo[0].fill_(1.0)
    a, sa = po.run_block(xo, params=params)
    b, sb = po.run_block(xo, params=params, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert torch.equal(a, b) and torch.equal(sa, sb)
    io = np.array([0, 1, 777, no - 1])
    xh = np.zeros((1024, len(io), 1), np.float32)
    xh[0] = 1.0
    assert ndiff(a[:, torch.as_tensor(io, device="cuda")].cpu().numpy(), C.osc_chain(np.ascontiguousarray(W.osc_chain_params(SEED + 1, io)), xh)) == 0


@pytest.mark.parametrize("seed", range(24))
def test_wave_split_random_cascades(torch_cuda, F, seed):
    """random serial filters (form, 4 .. 16 stages, random stable coefficients), random stream counts and block lengths,
    every split the graph allows with a random unroll and workgroup size -- against the oracle, two chained blocks."""
    rng = np.random.default_rng(9000 + seed)
    form = ["df1", "df2", "df1t"][seed % 3]
    n = int(rng.choice([4, 6, 8, 10, 12, 16]))

    def stage():
        if form == "df1t":
            return G.df1t()
        r, th = rng.uniform(0.3, 0.95), rng.uniform(0.1, 3.0)                      # poles inside the unit circle
        c = (rng.uniform(0.1, 1.0), rng.uniform(-1, 1), rng.uniform(-1, 1), 2 * r * np.cos(th), -r * r)
        return (G.df1 if form == "df1" else G.df2)(*[float(np.float32(v)) for v in c])

    g = stage()
    for _ in range(n - 1):
        g = G.seq(g, stage())
    prog = F.compile(F.from_sexpr(g))
    ns, T = int(rng.integers(1, 400)), int(rng.integers(1, 700))
    x = O.synth_input(seed, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert ndiff(ref, want) == 0
    tried = 0
    for W in (1, 2, 3, 4):
        io = F.C.FZ_VF_IO_WAVE if (W == 1 or rng.random() < 0.5) else 0
        v = F.make_variant(1, int(rng.choice([8, 16, 32])), int(rng.choice([0, 64, 128])), F.C.FZ_VF_WAVES(W) | io)
        try:
            prog.kernel_name(v, ns, T)
        except F.FlowzError:
            continue
        cut = int(rng.integers(0, T + 1))
        a, st = run_gpu(torch_cuda, F, prog, x[:cut], variant=v) if cut else (x[:0], None)
        b, st = run_gpu(torch_cuda, F, prog, x[cut:], variant=v, state=st) if cut < T else (x[:0], st)
        assert ndiff(np.concatenate([a, b]), want) == 0, (form, n, ns, T, W, cut)
        assert ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0
        tried += 1
    assert tried >= 1


@pytest.mark.parametrize("T", [5, 64, 300, 1000])
def test_wave_split_with_scalar_prefix_and_per_stream_coefficients(torch_cuda, F, T):
    """config 4's graph -- resonator (scalar prefix) -> 6 DF1 stages, 31 per-stream coefficients -- in two and three parts, with and
    without an I/O wave: the prefix and its coefficients go with part 0; against the compiled C oracle, chained blocks."""
    ns = 200
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    P = W.osc_chain_params(SEED + 3, np.arange(ns))
    x = np.zeros((T, ns, 1), np.float32)
    x[0] = 1.0
    want = C.osc_chain(P, x)
    ref, st_ref = run_gpu(torch_cuda, F, prog, x, params=P, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    assert ndiff(ref, want) == 0
    for fl in (F.C.FZ_VF_WAVES(2), F.C.FZ_VF_WAVES(3), F.C.FZ_VF_WAVES(2) | F.C.FZ_VF_IO_WAVE, F.C.FZ_VF_WAVES(3) | F.C.FZ_VF_IO_WAVE):
        v = F.make_variant(1, 16, 0, fl)
        got, st = run_gpu(torch_cuda, F, prog, x, params=P, variant=v)
        assert ndiff(got, want) == 0 and ndiff(st.cpu().numpy(), st_ref.cpu().numpy()) == 0, (T, fl)
        cut = T // 3 + 1
        a, st1 = run_gpu(torch_cuda, F, prog, x[:cut], params=P, variant=v)
        b, _ = run_gpu(torch_cuda, F, prog, x[cut:], params=P, variant=v, state=st1) if cut < T else (x[:0], None)
        assert ndiff(np.concatenate([a, b]), want) == 0, (T, fl)
    with pytest.raises(F.FlowzError):
        prog.kernel_name(F.make_variant(1, 16, 0, F.C.FZ_VF_WAVES(4)), ns, T)         # six segments do not make four parts


def test_wave_split_windows_and_block_rate_coefficients(torch_cuda, F):
    """windows of a long recording (fz_run_block_window: rows [row0, row0 + n) of larger frame buffers, time-major and tiled) through
    the wave-split and I/O-wave kernels, per-stream coefficients swapped between the windows (config 4's graph)."""
    torch = torch_cuda
    ns, T, tile = 1024, 900, 256
    g = G.osc_chain(6)
    prog = F.compile(F.from_sexpr(g))
    x = O.synth_input(SEED + 93, np.arange(ns), T)
    cuts = [0, 300, 301, 640, 900]
    Ps = [W.osc_chain_params(SEED + 94 + k, np.arange(ns)) for k in range(len(cuts) - 1)]
    f = O.compile(g, ns, params=Ps[0])
    want = []
    for k in range(len(cuts) - 1):
        f._params = np.ascontiguousarray(Ps[k], np.float32)
        want.append(f.run(x[cuts[k]:cuts[k + 1]]))
    want = np.concatenate(want)
    variants = [F.make_variant(1, 16, 0, F.C.FZ_VF_WAVES(2)), F.make_variant(1, 32, 0, F.C.FZ_VF_WAVES(3) | F.C.FZ_VF_IO_WAVE),
                F.make_variant(1, 16, 0, F.C.FZ_VF_IO_WAVE), F.make_variant(1, 8, 64, F.C.FZ_VF_WAVES(3))]
    for tiled in (False, True):
        xd = torch.from_numpy(x).cuda()
        xd = F.to_tiled(xd, tile) if tiled else xd
        od = torch.zeros_like(xd)
        st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
        for k in range(len(cuts) - 1):
            prog.run_window(xd, od, st, cuts[k], cuts[k + 1] - cuts[k], params=torch.from_numpy(Ps[k]).cuda(), variant=variants[k])
        got = (F.from_tiled(od) if tiled else od).contiguous().cpu().numpy()
        assert ndiff(got, want) == 0, tiled


def test_wave_split_blocks_chain_tiles_and_refusals(torch_cuda, F):
    """blocks of a wave-split kernel chain with every other variant through the canonical state; stream-tiled frames;
    graphs that are not two isomorphic halves are refused."""
    torch = torch_cuda
    g = G.df1_cascade(6)
    prog = F.compile(F.from_sexpr(g))
    ws = F.make_variant(1, 16, 0, WAVE_SPLIT)
    assert prog.kernel_name(ws, 4096, 256).startswith("fz_block_kernel_p1u16b128w2f")
    ns = 4096 + 64 * 3
    x = O.synth_input(11, np.arange(ns), 230)
    want = C.df1_cascade([G.STABLE] * 6, x)
    a, st = run_gpu(torch, F, prog, x[:33], variant=ws)
    b, st = run_gpu(torch, F, prog, x[33:34], variant=ws, state=st)                  # a 1-sample block
    c, st = run_gpu(torch, F, prog, x[34:70], variant=F.make_variant(1, 8, 256, STAGE_PACK), state=st)
    d, st = run_gpu(torch, F, prog, x[70:199], variant=ws, state=st)
    e, st = run_gpu(torch, F, prog, x[199:], variant=F.make_variant(2, 8), state=st)
    assert ndiff(np.concatenate([a, b, c, d, e]), want) == 0
    xt = torch.from_numpy(x[:, :4096]).cuda()
    yt, _ = prog.run_block(F.to_tiled(xt, 1024), variant=ws)                          # tiles of 1024 streams
    assert ndiff(F.from_tiled(yt).contiguous().cpu().numpy(), want[:, :4096]) == 0
    w3 = F.make_variant(1, 16, 0, F.C.FZ_VF_WAVES(3))                                # three parts of two biquads, one workgroup of three waves per 64 streams
    assert prog.kernel_name(w3, 4096, 256).startswith("fz_block_kernel_p1u16b64w3f")
    f, st = run_gpu(torch, F, prog, x[:101], variant=w3)
    h, st = run_gpu(torch, F, prog, x[101:], variant=ws, state=st)
    assert ndiff(np.concatenate([f, h]), want) == 0
    for bad in (G.df1_cascade(2), G.df1_cascade(3), G.par4_sum_fanout()):
        with pytest.raises(F.FlowzError):
            F.compile(F.from_sexpr(bad)).run_block(torch.zeros((4, 64, 1), device="cuda"), variant=ws)
    with pytest.raises(F.FlowzError):
        prog.run_block(torch.zeros((4, 64, 1), device="cuda"), variant=F.make_variant(2, 16, 64, WAVE_SPLIT))


# ---- empty blocks, maximum sizes -------------------------------------------------------------------------
def test_empty_block_is_a_noop(torch_cuda, F):
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1_cascade(2)))
    x = O.synth_input(1, np.arange(64), 10)
    y, st = run_gpu(torch, F, prog, x)
    before = st.clone()
    prog.run_block_ptr(None, y.ctypes.data if hasattr(y, "ctypes") else 0, st.data_ptr(), None, 64, 0)   # 0 samples
    prog.run_block_ptr(None, None, None, None, 0, 16)                                                      # 0 streams
    torch.cuda.synchronize()
    assert torch.equal(before, st)


def test_headline_workload_1M_x_4096_tiled_full_size(torch_cuda, F):
    """bench.py's workload on STREAM-TILED frames: 6-stage cascade, 1 048 576 streams x 4096 samples, tile 8192
    (16 GiB in, 16 GiB out).  Sampled streams (first/last tiles included) bitwise vs the compiled
    oracle over the full length; a second variant bit-identical on the whole output."""
    torch = torch_cuda
    ns, T, tile = 1 << 20, 4096, 8192
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((ns // tile, T, tile, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, st = prog.run_block(x)
    ids = _sample_ids(ns, 256, 5)
    xh = O.synth_input(SEED, ids, T)
    want = C.df1_cascade([G.STABLE] * 6, xh)
    tl, within = torch.from_numpy(ids // tile).cuda(), torch.from_numpy(ids % tile).cuda()
    got = y[tl, :, within, 0].T.cpu().numpy()                      # [T, n_ids]
    assert ndiff(got[:, :, None], want) == 0
    assert ndiff(x[tl, :, within, 0].T.cpu().numpy()[:, :, None], xh) == 0
    y2, st2 = prog.run_block(x, variant=F.make_variant(4, 4))
    assert torch.equal(y.view(torch.int32), y2.view(torch.int32))
    assert torch.equal(st.view(torch.int32), st2.view(torch.int32))


def test_headline_time_major_cascade6_1M_x_4096(torch_cuda, F, monkeypatch):
    """BASELINE's headline workload on its contract layout: the 6-stage cascade, 1 048 576 streams x 4096 samples, PLAIN time-major frames
    [t][stream] (rows 4 MiB apart), the library's static default -- the lockstep, XCD-synchronised row walk with four streams per lane.
    >= 1024 sampled streams (first / last included) bitwise against the compiled oracle over the full length; the whole output and the
    final state bit-identical to the free-running two-streams-per-lane kernel."""
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_AUTOTUNE", "0")                  # the static choice, whatever a first launch would measure on this board
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    ns, T = 1 << 20, 4096
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    assert prog.kernel_name(None, ns, T) == "fz_block_kernel_p4u1b1024f8912928"      # lockstep | XCD sync | three buffers of one row
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, st = prog.run_block(x)
    ids = _sample_ids(ns, 1024, 5)
    assert len(ids) >= 1024 and ids[0] == 0 and ids[-1] == ns - 1
    xh = O.synth_input(SEED, ids, T)
    want = C.df1_cascade([G.STABLE] * 6, xh)
    idt = torch.from_numpy(ids).cuda()
    assert ndiff(y[:, idt].cpu().numpy(), want) == 0
    assert ndiff(x[:, idt].cpu().numpy(), xh) == 0
    y2, st2 = prog.run_block(x, variant=F.make_variant(2, 16, 256))
    assert torch.equal(y.view(torch.int32), y2.view(torch.int32))
    assert torch.equal(st.view(torch.int32), st2.view(torch.int32))


def test_full_size_properties_power_of_two_scaling_and_time_shift(torch_cuda, F, monkeypatch):
    """Size-independent properties of the domain at BASELINE's full sizes, on EVERY stream and sample (the oracle checks sample streams): the graphs
    are linear and time-invariant, and in IEEE arithmetic two consequences hold bit for bit as long as nothing over- or underflows:
    scaling the input by a power of two scales every output by it (every product and sum is the same significand, another exponent), and an
    input delayed by d samples from zero state gives the output delayed by d samples.  Headline (6 x DF1, 1 M x 4096, plain rows, the lockstep
    default), config 3 (4-wire sum) and config 4 (oscillator chain, per-stream coefficients, dirac drive)."""
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    ns, T, d = 1 << 20, 4096, 7
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 5)
    y1, _ = prog.run_block(x)
    x.mul_(4.0)
    y2, _ = prog.run_block(x)
    y1.mul_(4.0)
    assert torch.equal(y1.view(torch.int32), y2.view(torch.int32))
    assert float(y2.abs().max()) < 16.0 and int((y2 != 0).sum()) > 0.99 * y2.numel()          # (a real signal, far from both ends of the exponent range)
    del y1
    xs = torch.empty_like(x)
    xs[:d].zero_()
    xs[d:].copy_(x[:-d])
    y3, _ = prog.run_block(xs)
    assert int((y3[:d] != 0).sum()) == 0 and torch.equal(y3[d:].view(torch.int32), y2[:-d].view(torch.int32))
    del x, xs, y2, y3
    torch.cuda.empty_cache()
    # config 3: four wires in, one out
    p4 = F.compile(F.from_sexpr(G.par4_sum()))
    x4 = torch.empty((T, ns, 4), dtype=torch.float32, device="cuda")
    F.synth_fill(x4, SEED + 6)
    a, _ = p4.run_block(x4)
    x4.mul_(0.5)
    b, _ = p4.run_block(x4)
    a.mul_(0.5)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    del x4, a, b
    torch.cuda.empty_cache()
    # config 4: the oscillator chain rings on a dirac; twice the dirac, twice the ringing, on every one of the million coefficient sets
    po = F.compile(F.from_sexpr(G.osc_chain(6)))
    params = torch.from_numpy(W.osc_chain_params(SEED + 1, np.arange(ns))).cuda()
    xo = torch.zeros((T, ns, 1), dtype=torch.float32, device="cuda")
    xo[0].fill_(1.0)
    a, _ = po.run_block(xo, params=params)
    xo[0].fill_(2.0)
    b, _ = po.run_block(xo, params=params)
    a.mul_(2.0)
    fin = torch.isfinite(b)
    assert bool(fin.all()) and torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_remainder_launches_of_two_host_threads_keep_their_own_order(torch_cuda, F, monkeypatch):
    """A block of 262 145 streams runs as one lap of whole workgroups plus a REMAINDER launch on the program's side stream, forked from
    and joined to the caller's stream by events (fz_launch.cpp).  Two host threads, each on its own HIP stream, refill their input
    right before every launch: a remainder that waited on the OTHER thread's fork event would read the frames of the block before.
    (Round 4 shared the event pair between concurrent launches with no lock; the fork ... join sequence is now one critical section.)"""
    import threading
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_AUTOTUNE", "0")
    props = torch.cuda.get_device_properties(0)
    ns, T = props.multi_processor_count * 1024 + 1, 1024            # one stream more than the lap's workgroups hold
    prog = F.compile(F.from_sexpr(G.df1_cascade(2)))
    ids = np.array([0, 1, ns - 2, ns - 1])
    idt = torch.from_numpy(ids).cuda()
    bad, rounds = [], 6

    def worker(k):
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            x = torch.zeros((T, ns, 1), dtype=torch.float32, device="cuda")
            y = torch.empty_like(x)
            for r in range(rounds):
                seed = 1000 * k + r
                F.synth_fill(x, seed)                                # producer of THIS block on this thread's stream
                st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
                prog.run_block(x, state=st, out=y)
                got = y[:, idt].cpu().numpy()                       # (synchronises this stream only)
                want = C.df1_cascade([G.STABLE] * 2, O.synth_input(seed, ids, T))
                if ndiff(got, want):
                    bad.append((k, r, ndiff(got, want)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad


def test_cold_kernel_cache_first_launches_are_bounded(tmp_path):
    """A checkout without the pre-built code objects (zignal_amd/_kcache travels with a built tree, not with git): a fresh process with an
    EMPTY kernel cache builds and launches the library defaults of the headline (time-major 1 M x 4096), of config 2 (65 536 streams,
    time-major and tiled) and of the stream-major layouts of both -- every kernel through hiprtc -- within two minutes, and leaves them
    in the cache it was pointed at."""
    import subprocess
    import sys
    import time
    code = (
        "import torch, time\n"
        "from zignal_amd import flowz as F, workloads as W\n"
        "p = F.compile(F.from_sexpr(W.df1_cascade(6)))\n"
        "T = 4096\n"
        "for ns, tile, sm in ((1 << 20, 0, False), (65536, 0, False), (65536, 8192, False), (1 << 20, 0, True), (65536, 0, True)):\n"
        "    t0 = time.time()\n"
        "    if sm:\n"
        "        x = torch.zeros((ns, T, 1), dtype=torch.float32, device='cuda'); y, st = p.run_block_stream_major(x)\n"
        "    else:\n"
        "        x = torch.zeros((ns // tile, T, tile, 1) if tile else (T, ns, 1), dtype=torch.float32, device='cuda'); y, st = p.run_block(x)\n"
        "    torch.cuda.synchronize()\n"
        "    assert float(y.abs().max()) == 0.0\n"
        "    print('built+ran', ns, tile, sm, round(time.time() - t0, 1), flush=True)\n"
        "    del x, y, st\n")
    env = dict(os.environ, FLOWZ_HIP_CACHE=str(tmp_path), FLOWZ_HIP_AUTOTUNE="0", FLOWZ_HIP_NO_PLAN_CACHE="1")
    env.pop("FLOWZ_HIP_NO_CACHE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=os.path.dirname(HERE))
    wall = time.time() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    built = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(built) >= 4, built                                   # (config 2 runs one kernel on rows and on tiles)
    assert wall <= 120.0, (wall, out.stdout)


def test_many_streams_16M(torch_cuda, F):
    """16 777 216 streams x 24 samples: 64 MiB rows (time-major) and 2048 tiles; checks the 64-bit row /
    tile addressing at the far end of the buffers."""
    torch = torch_cuda
    ns, T = 1 << 24, 24
    prog = F.compile(F.from_sexpr(G.df1_cascade(2)))
    ids = np.unique(np.concatenate([[0, 1, ns - 1, ns - 2, ns - 8193, 12345678], np.random.default_rng(7).integers(0, ns, 64)]))
    xh = O.synth_input(SEED, ids, T)
    want = C.df1_cascade([G.STABLE] * 2, xh)
    idt = torch.from_numpy(ids).cuda()
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, _ = prog.run_block(x)
    assert ndiff(y[:, idt].cpu().numpy(), want) == 0
    xt = torch.empty((ns // 8192, T, 8192, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(xt, SEED)
    yt, _ = prog.run_block(xt)
    assert torch.equal(F.from_tiled(yt).contiguous().view(torch.int32), y.view(torch.int32))


def test_one_program_many_states_concurrently(torch_cuda, F):
    """A program handle is re-entrant for concurrent fz_run_block calls on different state buffers
    (SURVEY 8b): two host threads, two HIP streams, interleaved blocks."""
    import threading
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    ns, T, nblk = 4096, 64, 12
    xs = [O.synth_input(40 + k, np.arange(ns), T * nblk) for k in range(2)]
    want = [C.df1_cascade([G.STABLE] * 6, x) for x in xs]
    outs = [None, None]

    def worker(k):
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            xd = torch.from_numpy(xs[k]).cuda()
            st = torch.zeros((prog.n_state, ns), device="cuda")
            ys = []
            for b in range(nblk):
                y, st = prog.run_block(xd[b * T:(b + 1) * T].contiguous(), state=st,
                                       variant=F.make_variant(1 + k, 8))
                ys.append(y)
            stream.synchronize()
            outs[k] = torch.cat(ys).cpu().numpy()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(2):
        assert ndiff(outs[k], want[k]) == 0


def test_rbj_lowpass_coefficient_generator_feeds_stream_params(torch_cuda, F):
    """reactive_filter_coeff.cpp:38-58 on the device, written straight into the rows
    of the per-stream `params` buffer that a DF1 stage with fz_stream_param coefficients reads.
    Round 6: BIT parity with the checker.  sin / cos come from the same reduction + polynomial pair on both sides (IEEE double operations
    in a fixed order, no libm, no FMA), everything behind them is IEEE float / double arithmetic: no tolerance is left.  (Against the
    REFERENCE the generator stays unpinned -- reactive_expressions needs Boost --; its distance from glibc's sinf / cosf is bounded in
    tests/test_oracle_c.py.)"""
    torch = torch_cuda
    ns, T = 5000, 400
    rng = np.random.default_rng(11)
    freq = rng.uniform(30.0, 18000.0, ns).astype(np.float32)
    q = rng.uniform(0.4, 9.0, ns).astype(np.float32)
    raw6 = torch.empty((6, ns), device="cuda")
    params = torch.empty((5, ns), device="cuda")
    F.rbj_lowpass(torch.from_numpy(freq).cuda(), torch.from_numpy(q).cuda(), 44100.0, raw6=raw6, df1=params)
    torch.cuda.synchronize()
    want_raw, want_df1 = C.rbj_lowpass(freq, q, 44100.0)
    got_raw, got_df1 = raw6.cpu().numpy(), params.cpu().numpy()
    bad = np.flatnonzero((got_raw.view(np.uint32) != want_raw.view(np.uint32)).any(axis=0) | (got_df1.view(np.uint32) != want_df1.view(np.uint32)).any(axis=0))
    assert bad.size == 0, [(float(freq[i]).hex(), float(q[i]).hex(), got_raw[:, i].tolist(), want_raw[:, i].tolist()) for i in bad[:4]]
    # the whole audio band and beyond, other sample rates, w0 across several periods, tiny and huge Q: a million streams, still every bit
    nb = 1 << 20
    fb = np.concatenate([rng.uniform(0.0, 24000.0, nb // 2), 10.0 ** rng.uniform(-3.0, 6.5, nb // 2)]).astype(np.float32)
    qb = (10.0 ** rng.uniform(-3.0, 3.0, nb)).astype(np.float32)
    for sr in (44100.0, 48000.0, 8000.0):
        rb, pb = torch.empty((6, nb), device="cuda"), torch.empty((5, nb), device="cuda")
        F.rbj_lowpass(torch.from_numpy(fb).cuda(), torch.from_numpy(qb).cuda(), sr, raw6=rb, df1=pb)
        wr, wp = C.rbj_lowpass(fb, qb, sr)
        assert ndiff_nan_aware(rb.cpu().numpy(), wr) == 0 and ndiff_nan_aware(pb.cpu().numpy(), wp) == 0, sr
    # the filter itself, with the device-generated coefficients: bit-exact
    g = G.df1_param(0)
    prog = F.compile(F.from_sexpr(g))
    assert prog.n_param == 5
    x = O.synth_input(SEED + 4, np.arange(ns), T)
    y, _ = prog.run_block(torch.from_numpy(x).cuda(), params=params)
    want = O.compile(g, ns, params=got_df1).run(x)
    assert ndiff(y.cpu().numpy(), want) == 0
    assert np.isfinite(want).all() and np.abs(want).max() < 50.0      # low-pass: bounded response


# ---- delays beyond LDS: rings in HBM (the line's state rows), prefetched like extra wires -----------------
def far_graph():
    """feed-forward comb _1[_1000] into a feedback comb reading its own output 777 and 2 samples back"""
    return G.seq(G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 1000))),
                 G.fb(G.add(G.add(G.mul(G.lit(0.6), G.DEL(1, 777)), G.mul(G.lit(0.1), G.DEL(1, 2))), G.IN(2))))


@pytest.mark.parametrize("P", [1, 2, 4])
def test_far_delays_hbm_rings_chained_blocks(torch_cuda, F, P):
    g = far_graph()
    prog = F.compile(F.from_sexpr(g))
    assert prog.max_delay == 1000 and prog.n_lds_slots == 0 and prog.n_state == 1000 + 777 + 2
    ns = 140
    sizes = [5, 1, 300, 31, 1000, 777, 64, 16, 2000, 123]           # crosses both ring lengths several times
    x = O.synth_input(SEED + 6, np.arange(ns), sum(sizes))
    want = O.compile(g, ns).run(x)
    st, pos, outs = None, 0, []
    for k, n in enumerate(sizes):
        y, st = run_gpu(torch_cuda, F, prog, x[pos:pos + n], state=st, variant=F.make_variant(P if k % 2 == 0 else 1, 16 if k % 3 else 8))
        outs.append(y)
        pos += n
    assert ndiff(np.concatenate(outs), want) == 0
    assert np.abs(want[-500:]).max() > 1e-3                            # the combs are alive at the end


def test_far_delays_walk_the_rows_in_lockstep(torch_cuda, F, monkeypatch):
    """Round 5: graphs with delay lines in HBM rings take the lockstep / XCD-synchronised row walk on plain time-major rows too (chunks of
    at least two rows, reads prefetched a chunk ahead; no third buffer).  One wave per SIMD of work x 1100 rows: the library's choice by
    name, sampled streams against the oracle, the whole output and the final state (ring rows + phase) equal to the free-running kernel's;
    a second block shorter than the lockstep minimum continues the rings."""
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_AUTOTUNE", "0")
    ns = torch.cuda.get_device_properties(0).multi_processor_count * 1024
    L, GS = F.C.FZ_VF_LOCKSTEP, F.C.FZ_VF_GRID_SYNC
    for g in (W.far_comb(300) if hasattr(W, "far_comb") else None,
              G.fb(G.add(G.add(G.mul(G.lit(0.4), G.DEL(1, 300)), G.mul(G.lit(-0.3), G.DEL(1, 32))), G.add(G.mul(G.lit(0.1), G.DEL(1, 2)), G.IN(2))))):
        if g is None:
            from zignal_amd import workloads as ZW
            g = ZW.far_comb(300)
        prog = F.compile(F.from_sexpr(g))
        T = 1100
        name = prog.kernel_name(None, ns, T)
        assert name.endswith("f%d" % (L | GS)) and "b256" not in name, name
        x = torch.empty((T + 300, ns, 1), dtype=torch.float32, device="cuda")
        F.synth_fill(x, SEED + 9)
        st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
        y1, st = prog.run_block(x[:T], state=st)
        y2, st = prog.run_block(x[T:], state=st)                            # 300 rows: below the lockstep minimum, the free-running kernel goes on
        st_ref = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
        r1, st_ref = prog.run_block(x[:T], state=st_ref, variant=F.make_variant(1, 8, 256))
        r2, st_ref = prog.run_block(x[T:], state=st_ref, variant=F.make_variant(2, 16, 256))
        assert torch.equal(y1.view(torch.int32), r1.view(torch.int32)) and torch.equal(y2.view(torch.int32), r2.view(torch.int32))
        assert torch.equal(st.view(torch.int32), st_ref.view(torch.int32))
        ids = _sample_ids(ns, 96, 21)
        want = O.compile(g, len(ids)).run(O.synth_input(SEED + 9, ids, T + 300))
        idt = torch.from_numpy(ids).cuda()
        assert ndiff(torch.cat([y1, y2])[:, idt].cpu().numpy(), want) == 0


def test_far_delay_minimum_and_tiled_layout(torch_cuda, F):
    """smallest far-read distance (32 samples = two prefetch chunks) on a 300-deep line, tiled frames"""
    torch = torch_cuda
    g = G.fb(G.add(G.add(G.mul(G.lit(0.4), G.DEL(1, 300)), G.mul(G.lit(-0.3), G.DEL(1, 32))), G.IN(2)))
    prog = F.compile(F.from_sexpr(g))
    ns, T = 2048, 700
    x = O.synth_input(SEED + 7, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    xt = F.to_tiled(torch.from_numpy(x).cuda(), 512)
    yt, _ = prog.run_block(xt, variant=F.make_variant(2, 16))
    assert ndiff(F.from_tiled(yt).cpu().numpy(), want) == 0


@pytest.mark.parametrize("mid", [9, 12, 20, 31])
def test_far_delay_with_mid_range_reader(torch_cuda, F, mid):
    """A wire delayed beyond the LDS (HBM ring) that is ALSO read 9..31 samples back: the ring read is
    prefetched a chunk ahead, so the chunk shrinks to half the youngest read (unroll 4 for a 9-sample read)."""
    torch = torch_cuda
    g = G.seq(G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))),
              G.add(G.add(G.mul(G.lit(0.25), G.DEL(1, 500)), G.mul(G.lit(-0.5), G.DEL(1, mid))), G.mul(G.lit(0.125), G.DEL(1, 3))))
    prog = F.compile(F.from_sexpr(g))
    ns, T = 300, 700
    x = O.synth_input(SEED + 77, np.arange(ns), T)
    want = O.compile(g, ns).run(x)
    for P in (1, 2, 4):
        got, _ = run_gpu(torch, F, prog, x, variant=F.make_variant(P, 0))
        assert ndiff(got, want) == 0, P
    xd = torch.from_numpy(x).cuda()
    ya, st = prog.run_block(xd[:333].contiguous())
    yb, _ = prog.run_block(xd[333:].contiguous(), state=st)
    assert ndiff(torch.cat([ya, yb]).cpu().numpy(), want) == 0
    with pytest.raises(F.FlowzError):
        prog.run_block(xd, variant=F.make_variant(1, 16))                  # the chunk must stay <= mid / 2


def test_large_graph_64_stages_320_coefficients(torch_cuda, F):
    """a 64-stage cascade with distinct coefficients: 576 ops per sample, 320 uniform coefficients in
    the kernarg segment, 130 state floats; plain, stage-packed (8 segments of 8 stages) and 2/lane."""
    rng = np.random.default_rng(0)
    coefs = [G.stable_biquad(rng.uniform(0.5, 0.9), rng.uniform(0.2, 2.8),
                             (rng.uniform(0.1, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2))) for _ in range(64)]
    prog = F.compile(F.from_sexpr(G.df1_cascade(64, coefs)))
    assert (prog.n_ops, prog.n_const, prog.n_state, prog.stage_packable) == (576, 320, 130, 1)
    ns, T = 130, 150
    x = O.synth_input(SEED + 8, np.arange(ns), T)
    want = C.df1_cascade(coefs, x)
    for v in (F.make_variant(1, 8, 256, NO_STAGE_PACK), F.make_variant(1, 8, 256, STAGE_PACK), F.make_variant(2, 4)):
        got, _ = run_gpu(torch_cuda, F, prog, x, variant=v)
        assert ndiff(got, want) == 0
    assert np.isfinite(want).all()


def test_tune_picks_a_plan_and_results_do_not_change(torch_cuda, F):
    """fz_program_tune: the measured plan is used by later launches without a variant; every candidate
    computes the same bits."""
    torch = torch_cuda
    ns, T = 1 << 18, 64
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((ns // 8192, T, 8192, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    ref, _ = prog.run_block(x, variant=F.make_variant(1, 8))
    chosen, ms = prog.tune(x)
    assert ms > 0 and chosen.streams_per_lane in (0, 1, 2, 4)
    got, _ = prog.run_block(x)                                   # planned variant
    assert torch.equal(got, ref)
    got2, _ = prog.run_block(x, variant=chosen)
    assert torch.equal(got2, ref)
    sample = _sample_ids(ns, 24, 5)
    xs = F.from_tiled(x)[:, sample].cpu().numpy()
    assert ndiff(F.from_tiled(got)[:, sample].cpu().numpy(), C.df1_cascade([G.STABLE] * 6, xs)) == 0
    # a few streams: the stage-packed candidates
    x2 = torch.empty((T, 4096, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x2, SEED + 1)
    r2, _ = prog.run_block(x2, variant=F.make_variant(1, 8, 256, F.C.FZ_VF_NO_STAGE_PACK))
    prog.tune(x2)
    g2, _ = prog.run_block(x2)
    assert torch.equal(g2, r2)


@pytest.mark.parametrize("pinned", [True, False])
def test_host_frames_pipelined_path(torch_cuda, F, pinned):
    """fz_bank_process_host on a long block: time chunks pipelined over three HIP streams (H2D, kernels,
    D2H), pinned or pageable host memory; equal to the device-resident path, state carried on."""
    torch = torch_cuda
    ns, T = 16384, 2304                                   # 144 MiB each way: 5 chunks of 512 steps (ragged last one)
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    xd = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(xd, SEED + 51)
    want, st = prog.run_block(xd)
    x = torch.empty((T, ns, 1), dtype=torch.float32, pin_memory=pinned)
    x.copy_(xd)
    bank = prog.bank(ns)
    y = bank.process_host(x if pinned else x.numpy())
    y = y if pinned else torch.from_numpy(y)
    assert torch.equal(y, want.cpu())
    # second block continues from the carried state (short: the single round-trip path)
    x2 = torch.empty((40, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x2, SEED + 52)
    want2, _ = prog.run_block(x2, state=st)
    y2 = bank.process_host(x2.cpu().numpy())
    assert np.array_equal(y2, want2.cpu().numpy())
    # float64 frames through the pipeline
    bank.reset()
    y64 = bank.process_host(x if pinned else x.numpy(), out_f64=True)
    y64 = y64 if pinned else torch.from_numpy(y64)
    assert torch.equal(y64, want.cpu().double())


def test_host_frames_pipelined_path_odd_stream_count(torch_cuda, F):
    """n_streams % 4 != 0 with rows above 512 KiB gives an odd chunk length (61 steps): the second
    pipeline slot must still start 16-byte aligned."""
    torch = torch_cuda
    ns, T = 137002, 200
    prog = F.compile(F.from_sexpr(G.df1_cascade(2)))
    xd = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(xd, SEED + 53)
    want, _ = prog.run_block(xd)
    bank = prog.bank(ns)
    y = bank.process_host(xd.cpu().numpy())
    assert np.array_equal(y.view(np.uint32), want.cpu().numpy().view(np.uint32))
    y64 = prog.bank(ns).process_host(xd.cpu().numpy(), out_f64=True)
    assert np.array_equal(y64, want.cpu().numpy().astype(np.float64))


def test_copy_probe_copies(torch_cuda, F):
    torch = torch_cuda
    for n in (4, 1020, 4096 + 8, 1 << 22):
        a = torch.randn(n, device="cuda")
        b = torch.zeros(n + 4, device="cuda")
        F.copy_probe(a, b[:n])
        torch.cuda.synchronize()
        assert torch.equal(a, b[:n]) and float(b[n:].abs().sum()) == 0.0


@pytest.mark.parametrize("T", [1, 15, 16, 17, 33, 48, 49, 80, 100])
def test_triple_buffered_prefetch_variant(torch_cuda, F, T):
    """FZ_VF_PREFETCH3 (loads two chunks ahead): every chunk-count remainder, plain and stage-packed."""
    ns = 640
    x = O.synth_input(SEED + 61, np.arange(ns), T)
    want = C.df1_cascade([G.STABLE] * 6, x)
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    P3 = F.C.FZ_VF_PREFETCH3
    for v in (F.make_variant(2, 16, 256, P3), F.make_variant(1, 16, 256, P3 | F.C.FZ_VF_STAGE_PACK), F.make_variant(4, 4, 256, P3),
              F.make_variant(1, 8, 256, P3 | F.C.FZ_VF_NO_STAGE_PACK)):
        got, _ = run_gpu(torch_cuda, F, prog, x, variant=v)
        assert ndiff(got, want) == 0


def test_block_launches_can_be_captured_in_a_hip_graph(torch_cuda, F):
    """fz_run_block only enqueues (no allocation, no synchronisation once the kernel is loaded): a chain of
    block launches captured in a hipGraph replays with identical results."""
    torch = torch_cuda
    ns, T, nblk = 4096, 48, 6
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    x = torch.empty((nblk, T, ns, 1), device="cuda")
    F.synth_fill(x.view(nblk * T, ns, 1), SEED + 71)
    y = torch.empty_like(x)
    st = torch.zeros((prog.n_state, ns), device="cuda")

    def blocks():
        for k in range(nblk):
            prog.run_block(x[k], state=st, out=y[k])
    blocks()
    torch.cuda.synchronize()
    ref, st_ref = y.clone(), st.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    st.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            blocks()
    st.zero_()
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and torch.equal(st, st_ref)
    want = C.df1_cascade([G.STABLE] * 6, x.view(nblk * T, ns, 1)[:, :64].cpu().numpy())
    assert ndiff(y.view(nblk * T, ns, 1)[:, :64].cpu().numpy(), want) == 0


@pytest.mark.parametrize("w", [1, 2, 3, 4, 7])
def test_stream_major_adapter(torch_cuda, F, w):
    """fz_transpose_frames: [stream][t][wire] <-> frames, time-major and tiled, ragged sizes."""
    torch = torch_cuda
    for ns, T, tile in ((64, 64, 0), (200, 130, 0), (1000, 33, 0), (4096, 70, 1024), (2048, 1100, 256), (5, 3, 0)):
        x = torch.randn((ns, T, w), device="cuda")
        fr = F.frames_from_stream_major(x, tile)
        want = x.permute(1, 0, 2).contiguous()                     # [T, ns, w]
        got = F.from_tiled(fr) if tile else fr
        assert torch.equal(got, want), (ns, T, tile)
        back = F.frames_to_stream_major(fr)
        assert torch.equal(back, x), (ns, T, tile)


def test_stream_major_buffers_end_to_end(torch_cuda, F):
    """One contiguous buffer per stream in (the reference's calling convention), the same out: adapter,
    block kernel, adapter -- against the compiled oracle in its stream-major layout."""
    torch = torch_cuda
    ns, T = 8192, 300
    xs = np.ascontiguousarray(np.transpose(O.synth_input(SEED + 81, np.arange(ns), T), (1, 0, 2)))     # [ns, T, 1]
    want = C.df1_cascade([G.STABLE] * 6, xs, stream_major=True)
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    fr = F.frames_from_stream_major(torch.from_numpy(xs).cuda(), 1024)
    y, _ = prog.run_block(fr)
    got = F.frames_to_stream_major(y).cpu().numpy()
    assert ndiff(got, want) == 0


def test_wide_rows_shrink_the_chunk_to_stay_below_4GiB(torch_cuda, F):
    """16 M streams x 4 input wires, time-major: 256 MiB rows.  A chunk of rows is addressed through one
    buffer descriptor (< 4 GiB), so the library lowers the unroll (16 -> 8 rows); a forced unroll fails."""
    torch = torch_cuda
    ns, T = 1 << 24, 20
    prog = F.compile(F.from_sexpr(G.par4_sum()))
    x = torch.empty((T, ns, 4), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    y, _ = prog.run_block(x)
    ids = np.unique(np.concatenate([[0, 1, ns - 1, ns - 2, 9999999], np.random.default_rng(11).integers(0, ns, 48)]))
    xh = O.synth_input(SEED, ids, T, n_wires=4)
    idt = torch.from_numpy(ids).cuda()
    assert ndiff(y[:, idt].cpu().numpy(), C.par4_sum(G.PAR4_SETS, xh)) == 0
    with pytest.raises(F.FlowzError):
        prog.run_block(x, variant=F.make_variant(1, 16))
    y8, _ = prog.run_block(x, variant=F.make_variant(1, 8))
    assert torch.equal(y8.view(torch.int32), y.view(torch.int32))


def test_block_windows_and_control_rate_coefficients(torch_cuda, F):
    """fz_run_block_window / fz_bank_process_blocks: a long (stream-tiled) recording processed in blocks,
    every block with its own per-stream coefficient set (std::ref modulation at block rate)."""
    torch = torch_cuda
    ns, T, L, tile = 2048, 200, 64, 512                      # 4 blocks: 64, 64, 64, 8
    nb = (T + L - 1) // L
    g = G.osc_chain(6)
    prog = F.compile(F.from_sexpr(g))
    x = O.synth_input(SEED + 91, np.arange(ns), T)
    Ps = [W.osc_chain_params(SEED + 92 + k, np.arange(ns)) for k in range(nb)]
    # oracle: one closure set, coefficients swapped between blocks
    f = O.compile(g, ns, params=Ps[0])
    want = []
    for k in range(nb):
        f._params = np.ascontiguousarray(Ps[k], np.float32)
        want.append(f.run(x[k * L:(k + 1) * L]))
    want = np.concatenate(want)
    xt = F.to_tiled(torch.from_numpy(x).cuda(), tile)
    pb = torch.from_numpy(np.stack(Ps)).cuda()
    out = torch.empty_like(xt)
    bank = prog.bank(ns)
    bank.process_blocks(xt, out, L, pb)
    assert ndiff(F.from_tiled(out).cpu().numpy(), want) == 0
    # the same through explicit windows on time-major buffers, mixed variants
    xd = torch.from_numpy(x).cuda()
    od = torch.empty_like(xd)
    st = torch.zeros((prog.n_state, ns), device="cuda")
    for k, v in zip(range(nb), (None, F.make_variant(2, 8), F.make_variant(1, 16), F.make_variant(4, 4))):
        prog.run_window(xd, od, st, k * L, min(L, T - k * L), params=pb[k], variant=v)
    assert ndiff(od.cpu().numpy(), want) == 0
    with pytest.raises(F.FlowzError):
        prog.run_window(xd, od, st, T - 10, 20, params=pb[0])          # window beyond the buffer


def test_lds_rings_take_the_row_walk_in_lockstep_on_plain_rows(torch_cuda, F):
    """Round 6: graphs with LDS rings on plain time-major rows of many streams run 256-lane workgroups in lockstep, XCD-synchronised, the
    resident workgroups as one lap of many (ahead of the free-running kernel on every one of six fresh allocations: profiles/r06/placement.txt).
    A stream count that fills neither the last workgroup nor the last lap, two chained blocks: the library's default by name, sampled streams
    against the oracle, the whole output and the state against the free-running kernel."""
    torch = torch_cuda
    g = G.lds_ring_comb()
    prog = F.compile(F.from_sexpr(g))
    ns, T = 300_000 + 37, 1100
    assert prog.kernel_name(None, ns, T) == "fz_block_kernel_p1u16b256f8912896M"          # lockstep | XCD step (rows off the 64-byte grid: merging stores)
    assert prog.kernel_name(None, 1 << 20, 4096) == "fz_block_kernel_p1u16b256f8912896" and prog.kernel_name(None, 1 << 20, 4096, 8192) == "fz_block_kernel_p1u32b256f0"
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 61)
    y1, st1 = prog.run_block(x[:600].contiguous())
    y2, st2 = prog.run_block(x[600:].contiguous(), state=st1.clone())
    r1, sr1 = prog.run_block(x[:600].contiguous(), variant=F.make_variant(1, 32, 256))
    r2, sr2 = prog.run_block(x[600:].contiguous(), state=sr1.clone(), variant=F.make_variant(1, 32, 256))
    assert torch.equal(y1, r1) and torch.equal(y2, r2) and torch.equal(st1, sr1) and torch.equal(st2, sr2)
    ids = np.concatenate([np.arange(3), np.random.default_rng(5).integers(0, ns, 90), np.arange(ns - 40, ns)])
    want = O.compile(g, len(ids)).run(O.synth_input(SEED + 61, ids, T))
    got = torch.cat([y1, y2])[:, torch.as_tensor(ids, device="cuda")].cpu().numpy()
    assert ndiff(got, want) == 0


def _cmp_graphs():
    return {"hard_clipper": G.hard_clipper(), "clipped_biquad": G.clipped_biquad(),
            "clipped_biquad_cascade": G.seq(G.clipped_biquad(), G.clipped_biquad(-0.25, 0.4), G.df1()),
            "logic": ("chan", ("chan", ("not", G.IN(1)), ("or", G.IN(1), G.lit(0.0))), ("mul", ("lit64", 2.0), ("lt", G.IN(1), ("lit64", 0.25)))),
            "compare_in_double": G.mul(("ge", G.mul(("lit64", 1.0000000001), G.IN(1)), G.IN(1)), G.IN(1)),
            "gated_feedback": G.fb(G.add(G.mul(G.mul(G.lit(0.9), G.DEL(1, 1)), ("lt", G.DEL(1, 1), G.lit(0.8))), G.IN(2))),
            "two_wire_select": G.add(G.mul(G.IN(1), ("gt", G.IN(1), G.IN(2))), G.mul(G.IN(2), ("le", G.IN(1), G.IN(2))))}


def _edge_input(seed, ns, T, n_wires=1):
    """noise with the values comparisons are sensitive to: +-0, the thresholds themselves, NaN, +-inf, denormals"""
    x = O.synth_input(seed, np.arange(ns), T, n_wires=n_wires)
    special = np.array([0.0, -0.0, 0.5, -0.5, 0.8, 0.25, np.nan, np.inf, -np.inf, 1e-40, -1e-40, np.float32(0.5) - np.float32(2 ** -25)], np.float32)
    rng = np.random.default_rng(seed)
    k = rng.integers(0, T, 16 * len(special)), rng.integers(0, ns, 16 * len(special)), rng.integers(0, n_wires, 16 * len(special))
    x[k] = np.tile(special, 16)
    return x


@pytest.mark.parametrize("name", sorted(_cmp_graphs()))
def test_comparison_and_logical_operators_on_gpu(torch_cuda, F, name):
    """SURVEY 8 row a4, widened in round 6: the comparison and logical operators of C++ in a flow-graph (a hard clipper and a biquad whose recursion
    runs through it, spelled with < <= > >= && only; !, ||, ==-style gating; a comparison decided in double).  Every lane packing, chained blocks,
    stream-major buffers, stream tiles -- against the oracle on inputs that hold +-0, the thresholds, NaN, +-inf and denormals; NaNs of any payload equal."""
    torch = torch_cuda
    g = _cmp_graphs()[name]
    prog = F.compile(F.from_sexpr(g))
    ns, T = 520, 132
    x = _edge_input(SEED + 77, ns, T, n_wires=max(prog.n_in, 1))
    with np.errstate(all="ignore"):
        want = O.compile(g, ns).run(x)
    xd = torch.from_numpy(x).cuda()
    for P in (1, 2, 4):
        y, st = prog.run_block(xd, variant=F.make_variant(P, 8))
        assert ndiff_nan_aware(y.cpu().numpy(), want) == 0, P
    ya, sta = prog.run_block(xd[:51].contiguous())
    yb, stb = prog.run_block(xd[51:].contiguous(), state=sta)
    assert ndiff_nan_aware(torch.cat([ya, yb]).cpu().numpy(), want) == 0
    ys, _ = prog.run_block_stream_major(xd.permute(1, 0, 2).contiguous())
    assert ndiff_nan_aware(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0
    yt, _ = prog.run_block(F.to_tiled(xd[:, :512].contiguous(), 128))
    assert ndiff_nan_aware(F.from_tiled(yt).cpu().numpy(), want[:, :512]) == 0


def test_clipped_biquad_at_scale_in_lockstep(torch_cuda, F):
    """the clipped biquad on 300 000 streams x 1100 samples of plain time-major rows: the library's default there is the CU-wide lockstep walk
    (graphs with comparisons are not stage-packed: lanes are packed instead); sampled streams against the oracle, everything against another variant"""
    torch = torch_cuda
    g = G.clipped_biquad()
    prog = F.compile(F.from_sexpr(g))
    ns, T = 300_000, 1100
    assert prog.kernel_name(None, ns, T).endswith("f%d" % (F.C.FZ_VF_LOCKSTEP | F.C.FZ_VF_GRID_SYNC)) and prog.stage_packable == 0
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 78)
    x.mul_(2.0)                                                   # (into the clipper)
    y, st = prog.run_block(x)
    r, sr = prog.run_block(x, variant=F.make_variant(1, 8))
    assert torch.equal(y, r) and torch.equal(st, sr)
    ids = np.concatenate([np.arange(4), np.random.default_rng(6).integers(0, ns, 120), np.arange(ns - 4, ns)])
    xi = (O.synth_input(SEED + 78, ids, T) * np.float32(2.0)).astype(np.float32)
    want = O.compile(g, len(ids)).run(xi)
    assert ndiff(y[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), want) == 0
    assert float(np.abs(want).max()) <= 0.5 and float((np.abs(want) == 0.5).mean()) > 0.01   # the clipper is at work


def test_tiles_walk_in_lockstep_for_light_graphs(torch_cuda, F):
    """Round 6: stream tiles of >= 2^19 streams whose tile a CU-wide workgroup of two streams per lane divides run their rows in lockstep, XCD-synchronised,
    two laps at 1 M streams: the library's default by name, sampled streams against the oracle, the whole output and the state against the free-running kernel,
    two chained blocks; a stream count of whole tiles that is not whole laps."""
    torch = torch_cuda
    g = G.df1_cascade(6)
    prog = F.compile(F.from_sexpr(g))
    tile, T = 8192, 1100
    ns = (1 << 19) + 3 * tile                                       # 67 tiles: two laps, the second far from full
    LG = F.C.FZ_VF_LOCKSTEP | F.C.FZ_VF_GRID_SYNC
    assert prog.kernel_name(None, ns, T, tile) == "fz_block_kernel_p2u2b1024f%d" % LG
    x = torch.empty((ns // tile, T, tile, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 91)
    y1, st1 = prog.run_block(x[:, :600].contiguous())
    y2, st2 = prog.run_block(x[:, 600:].contiguous(), state=st1.clone())
    v = F.make_variant(2, 16, 256, F.C.FZ_VF_MAX_WG(2))
    r1, sr1 = prog.run_block(x[:, :600].contiguous(), variant=v)
    r2, sr2 = prog.run_block(x[:, 600:].contiguous(), state=sr1.clone(), variant=v)
    assert torch.equal(y1, r1) and torch.equal(y2, r2) and torch.equal(st1, sr1) and torch.equal(st2, sr2)
    ids = np.concatenate([np.arange(3), np.random.default_rng(9).integers(0, ns, 100), np.arange(ns - 3, ns)])
    want = C.df1_cascade([G.STABLE] * 6, O.synth_input(SEED + 91, ids, T))
    idt = torch.as_tensor(ids, device="cuda")
    got = torch.cat([y1, y2], dim=1)[idt // tile, :, idt % tile, :].permute(1, 0, 2).contiguous().cpu().numpy()
    assert ndiff(got, want) == 0


SM_GRAPHS = {"cascade6": lambda: G.df1_cascade(6), "par4": G.par4_sum, "cross_wire": G.cross_wire, "integrator": G.integrator,
             "lds_ring": lambda: G.seq(G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 40))), G.fb(G.add(G.mul(G.lit(0.7), G.DEL(1, 23)), G.IN(2)))),
             "df2t": G.df2t}


@pytest.mark.parametrize("name", sorted(SM_GRAPHS))
def test_stream_major_kernel_vs_oracle(torch_cuda, F, name):
    """fz_run_block_stream_major: [stream][t][wire] buffers straight through the block kernel (LDS-transposed
    chunks), ragged stream counts, tails, windows, chained blocks -- vs the oracle."""
    torch = torch_cuda
    g = SM_GRAPHS[name]()
    prog = F.compile(F.from_sexpr(g))
    for ns, T in ((1, 4), (64, 32), (200, 100), (777, 68), (1000, 36)):
        x = O.synth_input(SEED + 95, np.arange(ns), T, n_wires=max(prog.n_in, 1))
        want = O.compile(g, ns).run(x)                                   # [T, ns, n_out]
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        y, st = prog.run_block_stream_major(xs)
        assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0, (ns, T)
        for v in ((2, 0), (2, 8), (1, 4), (1, 16)):                       # two streams per lane, other chunk depths
            if ns % v[0] == 0:
                yv, stv = prog.run_block_stream_major(xs, variant=F.make_variant(*v))
                assert torch.equal(yv, y) and torch.equal(stv, st), (ns, T, v)
        # same state as the frame kernel leaves
        _, st_ref = prog.run_block(torch.from_numpy(x).cuda())
        assert torch.equal(st, st_ref)
        # two windows of the same buffers, state carried (row0 multiple of 4)
        if T >= 36:
            out = torch.zeros_like(y)
            _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=20)
            prog.run_block_stream_major(xs, out=out, state=st2, row0=20)
            assert torch.equal(out, y), (ns, T)


def _wide_in_graphs():
    bq = [G.df1(*c) for c in G.PAR4_SETS]
    return {"par4_sum": G.par4_sum(),                                                                       # 4 wires in, 1 out: four chunks per out-run
            "par4_two_sums": G.seq(G.par(*bq), G.chan(G.add(G.IN(1), G.IN(2)), G.sub(G.IN(3), G.IN(4)))),   # 4 in, 2 out: two chunks per out-run
            "eight_wires": G.seq(G.par(*(bq + bq)), G.chan(G.add(G.add(G.IN(1), G.IN(2)), G.add(G.IN(3), G.IN(4))),
                                                            G.add(G.add(G.IN(5), G.IN(6)), G.add(G.IN(7), G.IN(8)))))}   # 8 in, 2 out


@pytest.mark.parametrize("name", ["par4_sum", "par4_two_sums", "eight_wires"])
def test_stream_major_wide_frames_hold_their_outputs(torch_cuda, F, name):
    """Stream-major buffers, wide frames in and narrow frames out (round 4): the outputs of n_in / n_out chunks wait in registers and
    leave as out-runs as long as the in-runs (FZ_SM_HOLD in the short-chunk body).  Whole runs, a shorter last run, tails, windows,
    ragged stream counts, every chunk depth -- against the oracle, and against the body that stores every chunk (two streams per lane)."""
    torch = torch_cuda
    g = _wide_in_graphs()[name]
    prog = F.compile(F.from_sexpr(g))
    for ns, T in ((64, 128), (130, 300), (777, 100), (1000, 260), (3, 516)):
        x = O.synth_input(SEED + 171, np.arange(ns), T, n_wires=prog.n_in)
        want = O.compile(g, ns).run(x)
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        y, st = prog.run_block_stream_major(xs)
        assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0, (ns, T)
        for v in ((1, 4), (1, 8), (1, 16), (1, 32), (2, 8)):
            if ns % v[0] == 0:
                try:
                    yv, stv = prog.run_block_stream_major(xs, variant=F.make_variant(*v))
                except F.FlowzError:
                    assert v[0] == 2 or name == "eight_wires"           # (patches that do not fit)
                    continue
                assert torch.equal(yv, y) and torch.equal(stv, st), (ns, T, v)
        # windows: the first 40 rows, then the rest (an out-run that starts off the run grid), state carried
        out = torch.zeros_like(y)
        _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=40)
        prog.run_block_stream_major(xs, out=out, state=st2, row0=40)
        assert torch.equal(out, y), (ns, T)


@pytest.mark.parametrize("T", [4, 8, 12, 36, 68, 160, 200, 264])
@pytest.mark.parametrize("name", ["cascade6", "cascade12_two_stages_per_segment", "cascade7_prefix_plus_6", "cascade4_smoothing_one_pole",
                                  "integrator_cascade4_gain", "df2_pair"])
def test_stream_major_stage_packed_kernel_vs_oracle(torch_cuda, F, name, T):
    """The stage-packed body of the stream-major kernel (segments skewed in time, outputs FZ_SKEW samples behind the
    inputs, output chunks completed in the next compute phase): every chunk depth, chunk-less blocks, ragged tails,
    ragged stream counts, windows and mixed-variant chains -- vs the oracle, 0 ULP, canonical state."""
    torch = torch_cuda
    g = PACKABLE[name]()
    prog = F.compile(F.from_sexpr(g))
    for ns in (1, 64, 333):
        x = O.synth_input(SEED + 96, np.arange(ns), T)
        want = O.compile(g, ns).run(x)                                   # [T, ns, 1]
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        _, st_ref = prog.run_block(torch.from_numpy(x).cuda(), variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
        skew = int(prog.source(F.make_variant(1, 16, 256, STAGE_PACK)).split("#define FZ_SKEW ")[1].split()[0])   # skewed units - 1
        for U in (8, 16, 32):
            v = F.make_variant(1, U, 0, STAGE_PACK)
            if U <= skew:                                                    # a chunk must be longer than the skew (6 biquads = 12 atoms: 11)
                with pytest.raises(F.FlowzError):
                    prog.run_block_stream_major(xs, variant=v)
                continue
            assert prog.kernel_name(F.make_variant(1, U, 0, STAGE_PACK | 128), ns, T).endswith("f136")          # ...s<K>f136: packed + stream-major
            y, st = prog.run_block_stream_major(xs, variant=v)
            assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0, (ns, T, U)
            assert torch.equal(st, st_ref), (ns, T, U)
        if T >= 36:                                                        # two windows, state carried, mixed bodies
            out = torch.zeros_like(y)
            _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=20, variant=F.make_variant(1, 8 if skew < 8 else 16, 0, STAGE_PACK))
            prog.run_block_stream_major(xs, out=out, state=st2, row0=20, variant=F.make_variant(1, 16, 0, NO_STAGE_PACK))
            assert torch.equal(out, y), (ns, T)
            out.zero_()
            _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=12, variant=F.make_variant(1, 8, 0, NO_STAGE_PACK))
            prog.run_block_stream_major(xs, out=out, state=st2, row0=12, variant=F.make_variant(1, 16, 0, STAGE_PACK))
            assert torch.equal(out, y), (ns, T)


def test_stream_major_stage_packing_is_automatic_for_long_blocks(torch_cuda, F):
    """From 16 x (skewed units - 1) samples on the stream-major kernel picks the stage-packed body by itself (any stream count);
    osc -> 6 DF1 with per-stream coefficients (scalar prefix) at a size where every wave of a block is busy."""
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    ns, T = 4096 + 78, 512
    assert "s6f" in prog.kernel_name(F.make_variant(0, 0, 0, 128), ns, T)
    assert "s6f" not in prog.kernel_name(F.make_variant(0, 0, 0, 128), ns, 72)
    P = W.osc_chain_params(SEED + 3, np.arange(ns))
    x = np.zeros((T, ns, 1), np.float32)
    x[0] = 1.0
    x[100] = -0.5
    xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
    y, st = prog.run_block_stream_major(xs, params=torch.from_numpy(P).cuda())
    assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), C.osc_chain(P, x)) == 0
    _, st_ref = prog.run_block(torch.from_numpy(x).cuda(), params=torch.from_numpy(P).cuda(), variant=F.make_variant(2, 8))
    assert torch.equal(st[:, :ns - 1], st_ref[:, :ns - 1]) and torch.equal(st, st_ref)


SM_LONG, SM_SHORT = 256, 512
LONG_GRAPHS = {"cascade6": lambda: G.df1_cascade(6), "cascade7_prefix_plus_6": PACKABLE["cascade7_prefix_plus_6"],
               "cascade4_smoothing_one_pole": PACKABLE["cascade4_smoothing_one_pole"], "df2_pair": PACKABLE["df2_pair"],
               "cascade12_two_stages_per_segment": PACKABLE["cascade12_two_stages_per_segment"],
               "df1": G.df1, "df2t": G.df2t, "integrator": G.integrator,
               "depth8_fir": lambda: G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 8)))}


@pytest.mark.parametrize("name", sorted(LONG_GRAPHS))
def test_stream_major_long_run_kernel_vs_oracle(torch_cuda, F, name):
    """The long-run body of the stream-major kernel (FZ_VF_SM_LONG: 512-byte runs per stream, in-place LDS patch, outputs
    taken 4 / 8 samples back in time under stage packing): automatic from 256 samples on, phases of 128 and 64 samples,
    ragged tails and stream counts, windows, chains with the short-chunk body -- vs the oracle, 0 ULP, canonical state."""
    torch = torch_cuda
    g = LONG_GRAPHS[name]()
    prog = F.compile(F.from_sexpr(g))
    for ns, T in ((333, 256), (64, 388), (1, 520), (777, 300)):
        x = O.synth_input(SEED + 98, np.arange(ns), T)
        want = O.compile(g, ns).run(x)
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        assert prog.kernel_name(F.make_variant(0, 0, 0, 128), ns, T).endswith(("f384", "f392"))      # 128 | 256 [| 8]: long-run body
        y, st = prog.run_block_stream_major(xs)                              # automatic: the long-run body
        assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0, (ns, T)
        _, st_ref = prog.run_block(torch.from_numpy(x).cuda(), variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
        assert torch.equal(st, st_ref), (ns, T)
        for v in (F.make_variant(1, 64, 0, SM_LONG), F.make_variant(1, 128, 0, SM_LONG | NO_STAGE_PACK), F.make_variant(1, 0, 64, SM_LONG)):
            yv, stv = prog.run_block_stream_major(xs, variant=v)
            assert torch.equal(yv, y) and torch.equal(stv, st), (ns, T, v.unroll, v.flags)
        # windows: long body, then the short-chunk body continues (and the other way round); row0 multiples of 4
        out = torch.zeros_like(y)
        cut = 132 if T > 260 else 128
        _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=cut, variant=F.make_variant(1, 64, 0, SM_LONG))
        prog.run_block_stream_major(xs, out=out, state=st2, row0=cut, variant=F.make_variant(0, 0, 0, SM_SHORT))
        assert torch.equal(out, y), (ns, T)
        out.zero_()
        _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=20, variant=F.make_variant(0, 0, 0, SM_SHORT))
        prog.run_block_stream_major(xs, out=out, state=st2, row0=20)
        assert torch.equal(out, y), (ns, T)


PAIR_GRAPHS = {"cascade6": lambda: G.df1_cascade(6), "cascade7_prefix_plus_6": PACKABLE["cascade7_prefix_plus_6"],
               "cascade12_two_stages_per_segment": PACKABLE["cascade12_two_stages_per_segment"],
               "df1": G.df1, "df2t": G.df2t, "integrator": G.integrator,
               "depth8_fir": lambda: G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 8)))}


@pytest.mark.parametrize("name", sorted(PAIR_GRAPHS))
def test_stream_major_pair_long_run_kernel_vs_oracle(torch_cuda, F, name):
    """The PAIR long-run body of the stream-major kernel (streams_per_lane = 2 with FZ_VF_SM_LONG: two streams per lane, every
    node one packed instruction, halves of 64 samples in an interleaved in-place patch, 256-byte in-runs, 512-byte out-runs
    held back in registers): ragged even stream counts (waves with idle lanes, a lone pair), blocks with and without full
    128-sample phases, ragged tails, windows and chains with the one-stream bodies -- vs the oracle, 0 ULP, canonical state."""
    torch = torch_cuda
    g = PAIR_GRAPHS[name]()
    prog = F.compile(F.from_sexpr(g))
    pair = F.make_variant(2, 64, 0, SM_LONG)
    # (a graph whose registers do not fit next to the 256 staging registers runs the one-stream long-run body instead: the 12-stage
    #  cascade does with the compiler bundled with the PyTorch wheel, not with the ROCm installation's)
    kn = prog.kernel_name(F.make_variant(2, 64, 0, SM_LONG | 128), 1024, 512)
    assert kn.startswith(("fz_block_kernel_p2u64b64f", "fz_block_kernel_p1u128b64")), kn
    if name in ("cascade6", "df1"):
        assert kn.startswith("fz_block_kernel_p2u64b64f"), kn
    for ns, T in ((334, 256), (2, 388), (778, 300), (130, 128), (64, 124), (1026, 640)):
        x = O.synth_input(SEED + 101, np.arange(ns), T)
        want = O.compile(g, ns).run(x)
        xs = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).cuda()
        y, st = prog.run_block_stream_major(xs, variant=pair)
        assert ndiff(y.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0, (ns, T)
        _, st_ref = prog.run_block(torch.from_numpy(x).cuda(), variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
        assert torch.equal(st, st_ref), (ns, T)
        yv, stv = prog.run_block_stream_major(xs, variant=F.make_variant(2, 0, 128, SM_LONG))       # two waves per workgroup
        assert torch.equal(yv, y) and torch.equal(stv, st), (ns, T)
        # windows: the pair body, then the library's own choice continues (and the other way round); row0 multiples of 4
        out = torch.zeros_like(y)
        cut = 132 if T > 260 else 64
        _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=cut, variant=pair)
        prog.run_block_stream_major(xs, out=out, state=st2, row0=cut)
        assert torch.equal(out, y), (ns, T)
        out.zero_()
        _, st2 = prog.run_block_stream_major(xs, out=out, n_samples=20, variant=F.make_variant(0, 0, 0, SM_SHORT))
        prog.run_block_stream_major(xs, out=out, state=st2, row0=20, variant=pair)
        assert torch.equal(out, y), (ns, T)
    with pytest.raises(F.FlowzError):                                            # an odd stream count has no pairs
        prog.run_block_stream_major(torch.zeros((7, 256, 1), device="cuda"), variant=pair)
    with pytest.raises(F.FlowzError):                                            # halves of 64 samples only
        prog.run_block_stream_major(torch.zeros((8, 256, 1), device="cuda"), variant=F.make_variant(2, 128, 0, SM_LONG))


def test_stream_major_pair_body_is_the_default_for_deep_graphs_on_many_streams(torch_cuda, F):
    """From 2^19 (even) streams on -- and from 2^17 on where its 512-stream workgroups fill the chip's rounds -- a deep 1-in/1-out graph with uniform coefficients runs the pair long-run body by itself
    (the 6-biquad cascade: 28.0 instructions per stream and step against 30.4 with stage packing); shallow graphs, fewer
    streams, odd counts and short blocks keep the one-stream bodies.  Full size: against the frame kernel on every stream,
    sampled streams against the C oracle."""
    torch = torch_cuda
    sm = F.make_variant(0, 0, 0, 128)
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    assert prog.kernel_name(sm, 1 << 20, 4096) == "fz_block_kernel_p2u64b64f384"
    assert prog.kernel_name(sm, 1 << 19, 256) == "fz_block_kernel_p2u64b64f384"
    assert prog.kernel_name(sm, 1 << 18, 4096).startswith("fz_block_kernel_p1u128b64s6f") and prog.kernel_name(sm, 1 << 17, 4096).startswith("fz_block_kernel_p1u128b64s6f")
    assert prog.kernel_name(sm, 3 << 16, 4096).startswith("fz_block_kernel_p1u128b64s6f")
    assert prog.kernel_name(sm, 1 << 16, 4096).startswith("fz_block_kernel_p1u128b64s6f")     # at most one wave per SIMD of work: one-wave workgroups
    assert prog.kernel_name(sm, (1 << 20) + 1, 4096).startswith("fz_block_kernel_p1u128b64s6f")
    assert prog.kernel_name(sm, 1 << 20, 128).startswith("fz_block_kernel_p1u")
    assert F.compile(F.from_sexpr(G.df1_cascade(2))).kernel_name(sm, 1 << 20, 4096).startswith("fz_block_kernel_p1u128b64f")
    assert F.compile(F.from_sexpr(G.osc_chain(6))).kernel_name(sm, 1 << 20, 4096) == "fz_block_kernel_p2u64b64f384"   # per-stream coefficients ride along as packed pairs (round 4)
    ns, T = (1 << 19) + 130, 644
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 7)
    yf, stf = prog.run_block(x)
    ys, sts = prog.run_block_stream_major(x.permute(1, 0, 2).contiguous())
    assert torch.equal(ys.permute(1, 0, 2).contiguous(), yf) and torch.equal(sts, stf)
    ids = _sample_ids(ns, 256, 9)
    want = C.df1_cascade([G.STABLE] * 6, O.synth_input(SEED + 7, ids, T))
    assert ndiff(yf[:, torch.from_numpy(ids).cuda()].cpu().numpy(), want) == 0


def test_stream_major_long_run_osc_chain_per_stream_coefficients_64k(torch_cuda, F):
    """resonator -> 6 DF1 with 31 per-stream coefficients (scalar prefix + 6 packed segments) through the long-run body at
    a size where every SIMD has a wave; the full output against the frame kernel, sampled streams against the C oracle."""
    torch = torch_cuda
    ns, T = 65536 + 64, 1024
    prog = F.compile(F.from_sexpr(G.osc_chain(6)))
    P = W.osc_chain_params(SEED + 3, np.arange(ns))
    pd = torch.from_numpy(P).cuda()
    x = torch.zeros((T, ns, 1), dtype=torch.float32, device="cuda")
    x[0] = 1.0
    x[500] = -0.25
    yf, stf = prog.run_block(x, params=pd)
    ys, sts = prog.run_block_stream_major(x.permute(1, 0, 2).contiguous(), params=pd)
    assert prog.kernel_name(F.make_variant(0, 0, 0, 128), ns, T).endswith("s6f392")
    assert torch.equal(ys.permute(1, 0, 2).contiguous(), yf) and torch.equal(sts, stf)
    ids = _sample_ids(ns, 256, 5)
    xh = np.zeros((T, len(ids), 1), np.float32)
    xh[0], xh[500] = 1.0, -0.25
    assert ndiff(yf[:, torch.from_numpy(ids).cuda()].cpu().numpy(), C.osc_chain(np.ascontiguousarray(P[:, ids]), xh)) == 0


def test_stream_major_kernel_rejects_what_it_cannot_do(torch_cuda, F):
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1_cascade(2)))
    with pytest.raises(F.FlowzError):
        prog.run_block_stream_major(torch.zeros((8, 30, 1), device="cuda"))                 # rows % 4 != 0
    with pytest.raises(F.FlowzError):
        prog.run_block_stream_major(torch.zeros((8, 32, 1), device="cuda"), variant=F.make_variant(4, 8))
    with pytest.raises(F.FlowzError):                          # 4-wire frames, two streams per lane, 32-sample chunks: one wave per CU
        F.compile(F.from_sexpr(G.par4_sum())).run_block_stream_major(torch.zeros((128, 64, 4), device="cuda"), variant=F.make_variant(2, 32))
    far = F.compile(F.from_sexpr(("seq", ("in", 1), ("add", ("in", 1), ("del", 1, 300)))))
    with pytest.raises(F.FlowzError):
        far.run_block_stream_major(torch.zeros((8, 32, 1), device="cuda"))


@pytest.mark.parametrize("pinned", [True, False])
def test_host_stream_major_buffers(torch_cuda, F, pinned):
    """fz_bank_process_host_stream_major: host rows per stream in, host rows per stream out (the reference's
    calling convention), pipelined 2-D copies + the stream-major kernel; state carried across calls."""
    torch = torch_cuda
    ns, T = 32768, 1100                                   # 2 chunks of 256 samples ... ragged last chunk
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    xd = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(xd, SEED + 97)
    want, st = prog.run_block(xd)
    xs = torch.empty((ns, T, 1), dtype=torch.float32, pin_memory=pinned)
    xs.copy_(xd.permute(1, 0, 2))
    bank = prog.bank(ns)
    y = bank.process_host_stream_major(xs if pinned else xs.numpy())
    y = y if pinned else torch.from_numpy(y)
    assert torch.equal(y.permute(1, 0, 2).contiguous(), want.cpu())
    x2 = torch.empty((37, ns, 1), dtype=torch.float32, device="cuda")        # a short odd block, carried state
    F.synth_fill(x2, SEED + 98)
    want2, _ = prog.run_block(x2, state=st)
    y2 = bank.process_host_stream_major(np.ascontiguousarray(x2.permute(1, 0, 2).cpu().numpy()))
    assert np.array_equal(np.transpose(y2, (1, 0, 2)), want2.cpu().numpy())


def test_sample_rate_modulators_through_the_chunked_host_paths(torch_cuda, F):
    """the host-frames pipelines cut a long block into time chunks and launch every chunk on a buffer of its own
    (row 0 of the buffer = sample t0 of the block); the modulator rows must follow the chunk, not start over at 0."""
    torch = torch_cuda
    ns, T = 16384, 2304                                   # 144 MiB of frames each way: 5 time chunks (ragged last one)
    g = G.modulated_mix()
    prog = F.compile(F.from_sexpr(g))
    rng = np.random.default_rng(15)
    md = torch.from_numpy(rng.uniform(-0.9, 0.9, (2, T)).astype(np.float32)).cuda()
    prog.set_modulation(md)
    xd = torch.empty((T, ns, prog.n_in), dtype=torch.float32, device="cuda")
    F.synth_fill(xd, SEED + 54)
    want, st = prog.run_block(xd)                          # one launch over the whole block
    ids = np.array([0, 1, 63, 64, 777, ns - 1])
    ref = O.compile(g, len(ids)).run(O.synth_input(SEED + 54, ids, T, n_wires=prog.n_in), mod=md.cpu().numpy())
    assert ndiff(want[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), ref) == 0
    y = prog.bank(ns).process_host(xd.cpu().numpy())
    assert np.array_equal(y.view(np.uint32), want.cpu().numpy().view(np.uint32))
    ys = prog.bank(ns).process_host_stream_major(np.ascontiguousarray(xd.permute(1, 0, 2).cpu().numpy()))
    assert np.array_equal(np.transpose(ys, (1, 0, 2)).view(np.uint32), want.cpu().numpy().view(np.uint32))
    with pytest.raises(F.FlowzError):
        prog.set_modulation(md.cpu())                      # (a host tensor is refused, not dereferenced)
    prog.set_modulation(md[:, :T - 8].contiguous())       # an array shorter than the block: refused by the chunk that would overrun it
    with pytest.raises(F.FlowzError):
        prog.bank(ns).process_host(xd.cpu().numpy())


@pytest.mark.parametrize("name", ["cascade2", "cross_wire", "osc_chain", "par4_sum"])
def test_time_major_lockstep_workgroups_vs_oracle(torch_cuda, F, name):
    """FZ_VF_LOCKSTEP (round 3): CU-wide workgroups of 1024 lanes that meet at a barrier after every chunk -- the library's
    choice for plain time-major frames of many streams.  Ragged stream counts (a last workgroup with lanes AND whole waves
    missing: waves that have left do not count at the barrier), block lengths around the chunk sizes, three buffers, every
    lane packing; 0 ULP against the oracle, state included; blocks chain with the ordinary kernels."""
    torch = torch_cuda
    from zignal_amd import _capi
    L = _capi.FZ_VF_LOCKSTEP
    g = {"cascade2": lambda: G.df1_cascade(2), "cross_wire": G.cross_wire, "osc_chain": lambda: G.osc_chain(6), "par4_sum": G.par4_sum}[name]()
    prog = F.compile(F.from_sexpr(g))
    for ns, T in ((1024 * 4 + 260, 37), (3000, 64), (5000, 1), (2048, 130)):
        x = O.synth_input(SEED + 31, np.arange(ns), T, n_wires=max(prog.n_in, 1))
        params = W.osc_chain_params(SEED + 32, np.arange(ns)) if prog.n_param else None
        want = C.osc_chain(params, x) if params is not None else O.compile(g, ns).run(x)
        got0, st0 = run_gpu(torch, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK), params=params)
        assert ndiff(got0, want) == 0
        GS = _capi.FZ_VF_GRID_SYNC                                   # + the workgroups of an XCD synchronised through counters in memory
        for v in ((1, 8, 1024, L), (2, 2, 1024, L), (4, 1, 1024, L | _capi.FZ_VF_PREFETCH3), (2, 4, 512, L), (1, 16, 256, L | NO_STAGE_PACK),
                  (1, 4, 1024, L | GS), (2, 2, 1024, L | GS), (4, 1, 1024, L | GS | _capi.FZ_VF_PREFETCH3), (1, 8, 64, L | GS)):
            if ns % v[0] or prog.kernel_resources(F.make_variant(*v), ns, T)["scratch_bytes"]:
                continue                                         # (a register-heavy graph does not fit 128 registers per lane: the library never picks that)
            got, st = run_gpu(torch, F, prog, x, variant=F.make_variant(*v), params=params)
            assert ndiff(got, want) == 0 and torch.equal(st, st0), (name, ns, T, v)
        # two blocks, the second one by an ordinary kernel from the lockstep kernel's state
        if T >= 37 and ns % 2 == 0:
            xd = torch.from_numpy(x).cuda()
            pd = torch.from_numpy(params).cuda() if params is not None else None
            out = torch.zeros((T, ns, prog.n_out), device="cuda")
            st = torch.zeros((max(prog.n_state, 1), ns), device="cuda")
            prog.run_window(xd, out, st, 0, 20, params=pd, variant=F.make_variant(2, 2, 1024, L))
            prog.run_window(xd, out, st, 20, T - 20, params=pd, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
            assert ndiff(out.cpu().numpy(), want) == 0 and torch.equal(st, st0)


def test_time_major_default_is_lockstep_and_equals_the_plain_kernel(torch_cuda, F, monkeypatch):
    """Plain time-major frames from 262 144 streams on: the library picks the lockstep workgroups by itself (one, two, four
    streams per lane as the stream count allows; a register-heavy graph steps down) -- same bits as the four-wave workgroups;
    tiled frames and 4-wire frames keep theirs."""
    torch = torch_cuda
    from zignal_amd import _capi
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    LG = _capi.FZ_VF_LOCKSTEP | _capi.FZ_VF_GRID_SYNC
    # streams per lane, lanes per workgroup and laps follow from the CU count (time_major_geometry): four streams per lane
    # where their kernel fits the registers a lane of that workgroup gets (any host process builds with the installation's compiler
    # now: fz_rtc_worker), else the library steps down
    fits4 = prog.kernel_resources(F.make_variant(4, 1, 1024, LG | _capi.FZ_VF_PREFETCH3), 1 << 20, 4096, as_launched=False)["scratch_bytes"] == 0
    assert fits4
    assert prog.kernel_name(None, 1 << 20, 4096, 0) == "fz_block_kernel_p4u1b1024f%d" % (LG | _capi.FZ_VF_PREFETCH3)
    assert prog.kernel_name(None, 1 << 19, 4096, 0) == "fz_block_kernel_p2u2b1024f%d" % LG
    assert prog.kernel_name(None, 1 << 18, 4096, 0) == "fz_block_kernel_p2u4b512f%d" % LG          # two streams per lane, 512-lane workgroups: 0.74 against 0.65 for one x 1024
    assert prog.kernel_name(None, 3 << 18, 4096, 0) == "fz_block_kernel_p4u1b768f%d" % (LG | _capi.FZ_VF_PREFETCH3)   # 786 432 = 256 workgroups x 768 lanes x 4
    assert prog.kernel_name(None, 1 << 21, 4096, 0) == "fz_block_kernel_p2u2b1024f%d" % LG          # four laps, one launch each (no persistent kernel any more)
    # tiles of light graphs walk in lockstep too since round 6 (ahead on three boards: profiles/r06/tiles_in_lockstep.txt); short blocks, small tiles, the
    # oscillator chain's 31 coefficients per stream: two free-running workgroups per CU
    assert prog.kernel_name(None, 1 << 20, 4096, 8192) == "fz_block_kernel_p2u2b1024f%d" % LG
    assert prog.kernel_name(None, 1 << 20, 512, 8192) == prog.kernel_name(None, 1 << 20, 4096, 1024) == "fz_block_kernel_p2u16b256f%d" % _capi.FZ_VF_MAX_WG(2)
    assert F.compile(F.from_sexpr(G.osc_chain(6))).kernel_name(None, 1 << 20, 4096, 8192) == "fz_block_kernel_p2u16b256f%d" % _capi.FZ_VF_MAX_WG(2)
    assert "b1024" not in prog.kernel_name(None, 1 << 17, 4096, 0)
    # a register-heavy graph steps down: with many per-stream coefficients (the oscillator chain: 31) straight to one stream per lane,
    # stage-packed (packing by stages costs no registers per stream)
    assert F.compile(F.from_sexpr(G.osc_chain(6))).kernel_name(None, 1 << 20, 4096, 0) == "fz_block_kernel_p1u4b1024s6f%d" % (LG | _capi.FZ_VF_STAGE_PACK)
    assert F.compile(F.from_sexpr(G.osc_chain(8))).kernel_name(None, 1 << 20, 4096, 0) == "fz_block_kernel_p1u4b1024s8f%d" % (LG | _capi.FZ_VF_STAGE_PACK)
    # wide frames (round 4): one stream per lane in 1024-lane workgroups; the default on 4-wire frames equals the four-wave workgroups, laps and remainder included
    p4 = F.compile(F.from_sexpr(G.par4_sum()))
    assert p4.kernel_name(None, 1 << 20, 4096, 0) == "fz_block_kernel_p1u3b1024f%d" % LG and p4.kernel_name(None, 1 << 18, 4096, 0) == "fz_block_kernel_p1u1b1024f%d" % (LG | _capi.FZ_VF_PREFETCH3)
    ns, T = (1 << 18) + 1024 + 5, 1030
    x4 = torch.empty((T, ns, 4), dtype=torch.float32, device="cuda")
    F.synth_fill(x4, SEED + 35)
    y4, s4 = p4.run_block(x4)
    y40, s40 = p4.run_block(x4, variant=F.make_variant(1, 16, 256))
    assert torch.equal(y4, y40) and torch.equal(s4, s40)
    ids = np.array([0, 63, 1024, 262143, 262144, ns - 1])
    assert ndiff(y4[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), O.compile(G.par4_sum(), len(ids)).run(O.synth_input(SEED + 35, ids, T, n_wires=4))) == 0
    del x4, y4, y40
    # more blocks than the chip holds workgroups (300 of 1024 lanes): two laps, a launch each with counters of its own
    ns, T = 300 * 1024 + 64, 40
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 34)
    yg, stg = prog.run_block(x, variant=F.make_variant(1, 4, 1024, LG))
    y0, st0 = prog.run_block(x, variant=F.make_variant(2, 16, 256))
    assert torch.equal(yg, y0) and torch.equal(stg, st0)
    monkeypatch.setenv("FLOWZ_HIP_AUTOTUNE", "0")
    ns, T = (1 << 18) + 8, 1030                                 # ragged on purpose (whole laps + a remainder launch of 8 streams)
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 33)
    y, st = prog.run_block(x)
    y0, st0 = prog.run_block(x, variant=F.make_variant(2, 16, 256))
    assert torch.equal(y, y0) and torch.equal(st, st0)
    ids = np.array([0, 1, 63, 64, 1023, 1024, ns - 1])
    want = C.df1_cascade([G.STABLE] * 6, O.synth_input(SEED + 33, ids, T))
    assert ndiff(y[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), want) == 0


@pytest.mark.parametrize("ns", [1027, 2050, 4101, 1023])
def test_ragged_stream_counts_in_lockstep_workgroups(torch_cuda, F, ns):
    """FZ_VF_RAGGED (round 4): a stream count that is not a multiple of the streams per lane in the lockstep frame kernel.  The last lane's
    accesses run past the end of every row (rows of an odd count are only 4-byte aligned): the per-row buffer descriptors return zeros
    there and drop the writes (raw buffers are range-checked per dword, tools/oob_probe.hip).  Frames, state rows, per-stream
    coefficient rows; vs the oracle and the ordinary kernel, state included; chained blocks."""
    torch = torch_cuda
    from zignal_amd import _capi
    L, GS, P3 = _capi.FZ_VF_LOCKSTEP, _capi.FZ_VF_GRID_SYNC, _capi.FZ_VF_PREFETCH3
    for g, with_params in ((G.df1_cascade(2), False), (G.cross_wire(), False), (G.osc_chain(2), True)):
        prog = F.compile(F.from_sexpr(g))
        T = 41
        x = O.synth_input(SEED + 131, np.arange(ns), T, n_wires=max(prog.n_in, 1))
        params = W.osc_chain_params(SEED + 132, np.arange(ns), 2) if with_params else None
        want = O.compile(g, ns, params=params).run(x)
        got0, st0 = run_gpu(torch, F, prog, x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK), params=params)
        assert ndiff(got0, want) == 0
        for v in ((4, 1, 256, L | GS | P3), (4, 2, 128, L), (2, 2, 256, L | GS), (2, 4, 64, L)):
            if ns % v[0] == 0:
                continue
            assert prog.kernel_name(F.make_variant(*v), ns, T).endswith("f%dRM" % v[3])   # (internal bits as letters: R = rows clipped per descriptor, M = merging stores for rows off the 64-byte grid)
            got, st = run_gpu(torch, F, prog, x, variant=F.make_variant(*v), params=params)
            assert ndiff(got, want) == 0 and torch.equal(st, st0), (ns, v)
            a, st1 = run_gpu(torch, F, prog, x[:17], variant=F.make_variant(*v), params=params)
            b, st2 = run_gpu(torch, F, prog, x[17:], variant=F.make_variant(1, 8, 256, NO_STAGE_PACK), params=params, state=st1)
            assert ndiff(np.concatenate([a, b]), want) == 0 and torch.equal(st2, st0)
    with pytest.raises(F.FlowzError):                            # only the lockstep frame kernel on plain time-major rows takes such counts
        F.compile(F.from_sexpr(G.df1_cascade(2))).run_block(torch.zeros((8, 1027, 1), device="cuda"), variant=F.make_variant(2, 8, 256))


@pytest.mark.parametrize("ns", [(1 << 18) + 1, (1 << 18) + 515, 3 * (1 << 18) + 2, (1 << 20) + 1, 3 << 19, 1_000_001])
def test_time_major_laps_remainders_and_ragged_defaults(torch_cuda, F, ns, monkeypatch):
    """What the library's own choice does with stream counts that are not whole workgroups x CUs (time_major_geometry): whole laps as
    launches of their own, the few streams beyond them as a remainder launch of the few-stream kernels, a count that is not a multiple
    of the streams per lane through FZ_VF_RAGGED.  Same bits as the ordinary kernel on every stream, state included; sampled streams
    (the first, the last, both sides of the main / remainder seam) against the oracle."""
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    monkeypatch.setenv("FLOWZ_HIP_AUTOTUNE", "0")               # the library's static choice is what is under test
    for g, with_params in ((G.df1_cascade(6), False), (G.osc_chain(6), True)):
        prog = F.compile(F.from_sexpr(g))
        T = 1024 + 7                                             # (the walk in lockstep is the library's choice from 1024 rows on)
        x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
        F.synth_fill(x, SEED + 140)
        pd = torch.from_numpy(W.osc_chain_params(SEED + 141, np.arange(ns))).cuda() if with_params else None
        y, st = prog.run_block(x, params=pd)
        y0, st0 = prog.run_block(x, params=pd, variant=F.make_variant(1, 16, 256, NO_STAGE_PACK))
        assert torch.equal(y, y0) and torch.equal(st, st0), (ns, with_params)
        seam = (ns // (1 << 18)) * (1 << 18)
        ids = np.unique(np.clip(np.array([0, 1, 1023, 1024, seam - 1, seam, seam + 1, ns - 2, ns - 1]), 0, ns - 1))
        xh = O.synth_input(SEED + 140, ids, T)
        want = C.osc_chain(np.ascontiguousarray(pd[:, torch.as_tensor(ids, device="cuda")].cpu().numpy()), xh) if with_params else C.df1_cascade([G.STABLE] * 6, xh)
        assert ndiff(y[:, torch.as_tensor(ids, device="cuda")].cpu().numpy(), want) == 0


def test_grid_sync_survives_a_busy_gpu_and_overlapping_streams(torch_cuda, F, monkeypatch):
    """FZ_VF_GRID_SYNC waits for workgroups that may not be running: (i) two synchronised launches of the same program on two
    streams at once (each stream has counters of its own; the CUs are shared, so neither launch has all its workgroups
    resident), (ii) a synchronised launch next to a long copy on another stream, (iii) inside a captured hipGraph.  The waits
    are bounded: everything finishes, and every bit equals the unsynchronised kernel's."""
    torch = torch_cuda
    from zignal_amd import _capi
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    LG = _capi.FZ_VF_LOCKSTEP | _capi.FZ_VF_GRID_SYNC
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    ns, T = 1 << 19, 256
    xs = [torch.empty((T, ns, 1), dtype=torch.float32, device="cuda") for _ in range(2)]
    for k, x in enumerate(xs):
        F.synth_fill(x, SEED + 40 + k)
    want = [prog.run_block(x, variant=F.make_variant(2, 16, 256))[0] for x in xs]
    v = F.make_variant(2, 2, 1024, LG)
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                prog.run_block(xs[k], out=outs[k], variant=v)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want[0]) and torch.equal(outs[1], want[1])
    big = torch.empty((1 << 28,), dtype=torch.float32, device="cuda")        # 1 GiB copies on the side
    side = torch.empty_like(big)
    outs[0].zero_()
    with torch.cuda.stream(streams[1]):
        for _ in range(4):
            side.copy_(big)
    with torch.cuda.stream(streams[0]):
        prog.run_block(xs[0], out=outs[0], variant=v)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want[0])
    # captured: the counters' reset is a memset node in front of the kernel node
    outs[1].zero_()
    st = torch.zeros((prog.n_state, ns), dtype=torch.float32, device="cuda")
    g = torch.cuda.CUDAGraph()
    prog.run_block(xs[1], state=st.clone(), out=outs[1], variant=v)            # (module loaded, counters allocated before the capture)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        prog.run_block(xs[1], state=st, out=outs[1], variant=v)
    outs[1].zero_()
    st.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[1], want[1])


def test_autotune_env_measures_the_plan_on_first_use(torch_cuda):
    """FLOWZ_HIP_AUTOTUNE=1 (opt-in since round 6): the first big block of a shape selects its plan by itself; state and results are
    what a plain launch gives (own process: the knob is read once per process).  The measurement is a one-off of bounded length:
    under two seconds for the first launch of the shape (kernels at hand only: nothing is JIT-compiled), an ordinary launch after it."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, graphs as G
from zignal_amd import flowz as F
prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
ns, T, tile = 1 << 18, 256, 8192
x = torch.empty((ns // tile, T, tile, 1), device="cuda"); F.synth_fill(x, 5)
ref, st_ref = prog.run_block(x, variant=F.make_variant(1, 8))
import time
torch.cuda.synchronize(); t0 = time.time()
y, st = prog.run_block(x)                     # measures, restores the state, then runs the block
torch.cuda.synchronize(); t_first = time.time() - t0
stc = st.clone(); torch.cuda.synchronize(); t0 = time.time()
y2, st2 = prog.run_block(x, state=stc)        # uses the remembered plan
torch.cuda.synchronize(); t_next = time.time() - t0
print(f"first launch of the shape {t_first * 1e3:.1f} ms, the next one {t_next * 1e3:.2f} ms")
assert t_first < 2.0 and t_next < 0.05         # the plan measurement is a bounded one-off: a >= 100 ms warm-up + two passes over the candidates at hand
r2, sr2 = prog.run_block(x, state=st_ref.clone(), variant=F.make_variant(1, 8))
assert torch.equal(y, ref) and torch.equal(st, st_ref) and torch.equal(y2, r2) and torch.equal(st2, sr2)
print("autotune ok")
'''
    env = dict(os.environ, FLOWZ_HIP_AUTOTUNE="1", FLOWZ_HIP_DEBUG="1", FLOWZ_HIP_NO_PLAN_CACHE="1")   # (a persisted plan would make the measurement unnecessary)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "autotune ok" in out.stdout, out.stdout + out.stderr[-2000:]
    assert out.stderr.count("[flowz_hip] tune ") >= 5                   # the candidates were measured
    # round 6: WITHOUT the variable a launch never measures anything by itself -- the same script runs the static plan from the first block on
    env.pop("FLOWZ_HIP_AUTOTUNE")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "autotune ok" in out.stdout, out.stdout + out.stderr[-2000:]
    assert out.stderr.count("[flowz_hip] tune ") == 0 and "[flowz_hip] launched" in out.stderr


def test_tuned_plan_is_persisted_per_graph_shape_and_board(torch_cuda, F, tmp_path, monkeypatch):
    """fz_program_tune's winner goes to <kernel cache>/plans.txt (graph structure, n_streams, tile, board UUID): a NEW
    program of the same structure -- other coefficient values -- launched without a variant picks it up; other shapes and
    FLOWZ_HIP_NO_PLAN_CACHE do not."""
    torch = torch_cuda
    monkeypatch.setenv("FLOWZ_HIP_CACHE", str(tmp_path))
    monkeypatch.delenv("FLOWZ_HIP_NO_PLAN_CACHE", raising=False)
    ns, T = 1 << 17, 256
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED)
    p1 = F.compile(F.from_sexpr(G.df1_cascade(2)))
    v0 = p1.plan(ns)
    assert (v0.streams_per_lane, v0.unroll, v0.block_threads, v0.flags) == (0, 0, 0, 0)
    chosen, _ = p1.tune(x)
    lines = (tmp_path / "plans.txt").read_text().splitlines()
    f = lines[0].split()                                   # "<format tag> <graph hash> <n_streams> <tile> <board> <P> <U> <block> <flags> <ms> <n_samples>"
    assert len(lines) == 1 and f[0] == "fzplan3" and f[2:4] == [str(ns), "0"] and f[-1] == str(T)
    (tmp_path / "plans.txt").write_text("0123456789abcdef 131072 0 deadbeef 2 8 256 0 0.1\n" + lines[0] + "\n")   # a line of an older format is not ours to read
    p2 = F.compile(F.from_sexpr(G.df1_cascade(2, [G.PAR4_SETS[0], G.PAR4_SETS[1]])))        # same structure, other coefficients
    got = p2.plan(ns)
    assert (got.streams_per_lane, got.unroll, got.block_threads, got.flags) == (chosen.streams_per_lane, chosen.unroll, chosen.block_threads, chosen.flags)
    y2, _ = p2.run_block(x)
    y1, _ = p2.run_block(x, variant=F.make_variant(1, 8))
    assert torch.equal(y1, y2)
    other = p2.plan(ns * 2)
    assert (other.streams_per_lane, other.unroll, other.block_threads, other.flags) == (0, 0, 0, 0)
    monkeypatch.setenv("FLOWZ_HIP_NO_PLAN_CACHE", "1")
    p3 = F.compile(F.from_sexpr(G.df1_cascade(2)))
    off = p3.plan(ns)
    assert (off.streams_per_lane, off.unroll, off.block_threads, off.flags) == (0, 0, 0, 0)


def test_in_place_blocks_when_frame_widths_match(torch_cuda, F):
    """`in == out` exactly is allowed when n_in == n_out (every kernel body reads a sample before
    it writes the slot): frame kernel (lane-packed, stage-packed), stream-major short and long bodies."""
    torch = torch_cuda
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    ns, T = 4096 + 64, 384
    x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
    F.synth_fill(x, SEED + 70)
    want, st_want = prog.run_block(x, variant=F.make_variant(1, 8, 256, NO_STAGE_PACK))
    for v in (F.make_variant(2, 16), F.make_variant(4, 8), F.make_variant(1, 16, 256, STAGE_PACK), F.make_variant(1, 32, 256, NO_STAGE_PACK)):
        buf = x.clone()
        y, st = prog.run_block(buf, out=buf, variant=v)
        assert torch.equal(buf, want) and torch.equal(st, st_want), (v.streams_per_lane, v.unroll, v.flags)
    xs = x.permute(1, 0, 2).contiguous()
    for v in (None, F.make_variant(0, 0, 0, SM_SHORT), F.make_variant(2, 16)):
        buf = xs.clone()
        y, st = prog.run_block_stream_major(buf, out=buf, variant=v)
        assert torch.equal(buf.permute(1, 0, 2).contiguous(), want) and torch.equal(st, st_want)


@pytest.mark.parametrize("P", [0, 1, 2, 4])
def test_sample_rate_modulators_on_gpu(torch_cuda, F, P):
    """fz_modulator (the std::ref terminal of flowz/README.md:42-61 re-read at SAMPLE rate inside a block): one value per
    sample for all streams, scalar loads in the kernel.  vs the oracle; every layout; windows (row0); equal to the reference's
    own protocol -- one call per sample with the referenced variable changed between calls (fz_uniform + per-sample blocks)."""
    torch = torch_cuda
    ns, T = 512, 96
    rng = np.random.default_rng(5)
    m = rng.uniform(-0.9, 0.9, (2, T)).astype(np.float32)
    md = torch.from_numpy(m).cuda()
    x = O.synth_input(SEED + 80, np.arange(ns), T)
    v = F.make_variant(P, 8) if P else None
    for g in (G.one_pole_modulated(), G.modulated_mix()):
        prog = F.compile(F.from_sexpr(g))
        want = O.compile(g, ns).run(x, mod=m)
        with pytest.raises(F.FlowzError):
            prog.run_block(torch.from_numpy(x).cuda(), variant=v)                    # no modulation array yet
        prog.set_modulation(md)
        got, st = run_gpu(torch, F, prog, x, variant=v)
        assert ndiff(got, want) == 0
        if P in (0, 1):
            yt, stt = prog.run_block(F.to_tiled(torch.from_numpy(x).cuda(), 256), variant=F.make_variant(1, 8, 64))
            assert ndiff(F.from_tiled(yt).contiguous().cpu().numpy(), want) == 0 and torch.equal(stt, st)
        # windows of the same buffers: the modulator array is indexed by the row of the buffer
        xd = torch.from_numpy(x).cuda()
        out = torch.zeros_like(xd)
        st2 = torch.zeros_like(st)
        prog.run_window(xd, out, st2, 0, 37, variant=v)
        prog.run_window(xd, out, st2, 37, T - 37, variant=v)
        assert ndiff(out.cpu().numpy(), want) == 0 and torch.equal(st2, st)
        if P in (0, 1, 2):
            ys, sts = prog.run_block_stream_major(xd.permute(1, 0, 2).contiguous(), variant=F.make_variant(P, 0) if P else None)
            assert ndiff(ys.permute(1, 0, 2).contiguous().cpu().numpy(), want) == 0 and torch.equal(sts, st)
    # the reference's protocol: per-sample calls, the referenced variable changed in between
    if P == 0:
        pu = F.compile(~(F.uniform(0, 0.0) * F._1[F._1] + F._2))
        pm = F.compile(F.from_sexpr(G.one_pole_modulated()))
        pm.set_modulation(md)
        xd = torch.from_numpy(x).cuda()
        ym, _ = pm.run_block(xd)
        stu = torch.zeros((1, ns), device="cuda")
        rows = []
        for t in range(T):
            pu.set_uniform(0, float(m[0, t]))
            yt, stu = pu.run_block(xd[t:t + 1].contiguous(), state=stu)
            rows.append(yt)
        assert torch.equal(torch.cat(rows), ym)
    # long blocks through the long-run stream-major body
    if P == 0:
        TL = 300
        ml = rng.uniform(-0.9, 0.9, (2, TL)).astype(np.float32)
        xl = O.synth_input(SEED + 81, np.arange(ns), TL)
        g = G.modulated_mix()
        prog = F.compile(F.from_sexpr(g))
        prog.set_modulation(torch.from_numpy(ml).cuda())
        ys, _ = prog.run_block_stream_major(torch.from_numpy(np.ascontiguousarray(np.transpose(xl, (1, 0, 2)))).cuda())
        assert ndiff(ys.permute(1, 0, 2).contiguous().cpu().numpy(), O.compile(g, ns).run(xl, mod=ml)) == 0
