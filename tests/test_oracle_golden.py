"""Pin the CPU oracle against the reference's own known answers and golden vectors."""
import json
import os

import numpy as np
import pytest

import graphs as G
from oracle import flowz_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "tests_cpp_known_answers.json")))
REF = json.load(open(os.path.join(HERE, "golden", "ref_biquad_vectors.json")))


def tup(x):
    return tuple(tup(v) for v in x) if isinstance(x, list) else x


def bits(hexlist):
    return np.array([int(h, 16) for h in hexlist], np.uint32).view(np.float32)


@pytest.mark.parametrize("case", KA["evaluation"] + KA["readme"], ids=lambda c: c["name"])
def test_tests_cpp_known_answers(case):
    f = O.compile(tup(case["graph"]))
    for ins, outs in case["calls"]:
        got = f.step(*[np.float32(v) for v in ins])
        assert [float(g[0]) for g in got] == [float(v) for v in outs]


@pytest.mark.parametrize("case", KA["arity"], ids=lambda c: c["name"])
def test_tests_cpp_arity(case):
    g = tup(case["graph"])
    assert O.input_arity(g) == case["ins"]
    assert O.output_arity(g) == case["outs"]
    if "max_input_delays" in case:
        assert list(O.max_input_delays(g)) == case["max_input_delays"]


def test_reference_coefficients():
    c = bits(REF["coeffs_b0_b1_b2_a1_a2"])
    assert [G.B0, G.B1, G.B2, G.A1, G.A2] == list(c)


FORMS = {"df1": G.df1, "df2": G.df2, "df1t": G.df1t, "df1x2": lambda: G.seq(G.df1(), G.df1()),
         "df1x6": lambda: G.seq(*[G.df1() for _ in range(6)])}       # six reference DF1 closures in series: the headline workload's shape


def ref_par4():
    """Four reference DF1 closures side by side on four input wires, summed left to right (config 3's shape)."""
    return G.seq(G.par(G.df1(), G.df1(), G.df1(), G.df1()), G.add(G.add(G.add(G.IN(1), G.IN(2)), G.IN(3)), G.IN(4)))


@pytest.mark.parametrize("drive", ["dirac", "noise"])
def test_par4_bitwise_vs_four_reference_lambdas(drive):
    """(bq|bq|bq|bq) |= (_1+_2+_3+_4) against a composition of the reference's own DF1 closures (oracle/ref_harness.inc: zref_par4)."""
    x = bits(REF["inputs4"][drive]).reshape(-1, 1, 4)
    got = O.compile(ref_par4()).run(x)[:, 0, 0]
    assert np.array_equal(got.view(np.uint32), bits(REF["outputs4"][drive]["par4"]).view(np.uint32))


@pytest.mark.parametrize("drive", ["dirac", "noise"])
@pytest.mark.parametrize("form", sorted(FORMS))
def test_biquad_forms_bitwise_vs_reference_lambdas(form, drive):
    """Flowz graphs DF1/DF2/DF1T (test/benchmark.cpp:32,62,87) are bit-equivalent to the
    reference's hand-written lambdas: 0 differing samples."""
    x = bits(REF["inputs"][drive])
    want = bits(REF["outputs"][drive][form])
    got = O.compile(FORMS[form]()).run(x[:, None])[:, 0, 0]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_df2t_graph_matches_lambda_only_on_dirac():
    """The DF2T lambda merges states (test/benchmark.cpp:116-126), so it equals the Flowz
    graph `fwdt |= ~bwdt` for the dirac drive only."""
    x = bits(REF["inputs"]["dirac"])
    got = O.compile(G.df2t()).run(x[:, None])[:, 0, 0]
    assert np.array_equal(got.view(np.uint32), bits(REF["outputs"]["dirac"]["df2t"]).view(np.uint32))
    xn = bits(REF["inputs"]["noise"])
    gn = O.compile(G.df2t()).run(xn[:, None])[:, 0, 0]
    ndiff = int((gn.view(np.uint32) != bits(REF["outputs"]["noise"]["df2t"]).view(np.uint32)).sum())
    assert ndiff > 0


@pytest.mark.parametrize("drive", ["dirac", "noise"])
def test_cross_wire_vs_x_wire(drive):
    x = bits(REF["inputs"][drive])
    y = O.compile(G.cross_wire()).run(x[:, None])
    assert np.array_equal(y[:, 0, 0].view(np.uint32), bits(REF["outputs"][drive]["xwire0"]).view(np.uint32))
    assert np.array_equal(y[:, 0, 1].view(np.uint32), bits(REF["outputs"][drive]["xwire1"]).view(np.uint32))


MODEL = json.load(open(os.path.join(HERE, "golden", "model_derived_vectors.json")))


@pytest.mark.parametrize("name", ["one_quad", "cross_wire_output1"])
def test_model_derived_vectors(name):
    """dirac responses computed by the survey's own model of flowz.hpp -- an independent reading of the
    evaluator, NOT reference output (the fixture says so).  one_quad and nested feedback have no reference-built vector at all."""
    want = np.array([float.fromhex(h) for h in MODEL[name]["h_hexfloat"]], np.float32)
    x = np.zeros((len(want), 1, 1), np.float32)
    x[0] = 1.0
    if name == "one_quad":
        got = O.compile(G.one_quad()).run(x)[:, 0, 0]
    else:
        got = O.compile(G.cross_wire()).run(x)[:, 0, 1]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)


def test_dirac_sums_like_sum_dirac():
    for form in ("df1", "df2", "df1t"):
        y = O.compile(FORMS[form]()).run(bits(REF["inputs"]["dirac"])[:, None])[:, 0, 0]
        s = np.float32(y[0])
        for v in y[1:]:
            s = np.float32(s + v)
        assert float(s).hex() == REF["dirac_sum_201"][form]


def test_streams_are_independent_and_vectorised():
    x = O.synth_input(7, [3, 11, 12], 64)
    f3 = O.compile(G.df1_cascade(3), 3).run(x)
    for j in range(3):
        fj = O.compile(G.df1_cascade(3), 1).run(x[:, j:j + 1])
        assert np.array_equal(f3[:, j].view(np.uint32), fj[:, 0].view(np.uint32))


def test_delay_free_loop_rejected():
    with pytest.raises(O.GraphError):
        O.compile(("fb", ("add", ("in", 1), ("in", 2)))).step(np.float32(1))


def test_synth_input_range_and_determinism():
    x = O.synth_input(20160512, np.arange(5), 1000)
    assert x.dtype == np.float32 and x.min() >= -1.0 and x.max() < 1.0
    assert np.array_equal(x, O.synth_input(20160512, np.arange(5), 1000))
    assert np.array_equal(x[10:20], O.synth_input(20160512, np.arange(5), 10, t0=10))


@pytest.mark.parametrize("case", KA["result_types"], ids=lambda c: "tests.cpp:" + c["lines"])
def test_result_types_float_double(case):
    """test_result_type_transform (tests.cpp:184-232), float/double cases: the oracle's evaluated types"""
    assert O.output_dtypes(tup(case["graph"])) == case["types"]
    # ... and with typed state (what ResultType itself computes, incl. tests.cpp:219 where a double goes THROUGH a delay line)
    assert O.output_dtypes_typed(tup(case["graph"])) == case.get("result_type", case["types"])
