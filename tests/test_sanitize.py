"""Host-side lowering and code generation under AddressSanitizer + UBSan on a few hundred graphs
(random ones, the typed ones, the BASELINE workloads and deliberately malformed ones)."""
import os
import shutil
import subprocess

import pytest

import graphs as G
import randgraphs as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "zignal_amd", "csrc")


def prefix(e):
    out = [str(e[0])]
    for c in e[1:]:
        out.append(prefix(c) if isinstance(c, tuple) else repr(float(c)) if isinstance(c, float) else str(int(c)))
    return " ".join(out)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_lowering_and_codegen_under_asan_ubsan(tmp_path):
    embed = os.path.join(ROOT, "zignal_amd", "lib", "fz_skeleton_embed.cpp")
    if not os.path.exists(embed):
        subprocess.check_call(["python3", os.path.join(CSRC, "embed.py"), os.path.join(CSRC, "fz_block_kernel.hip.inc"), embed])
    exe = str(tmp_path / "sanitize_lowering")
    srcs = [os.path.join(CSRC, f) for f in ("fz_expr.cpp", "fz_lower.cpp", "fz_split.cpp", "fz_codegen.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), "-I", CSRC, os.path.join(HERE, "cpp", "sanitize_lowering.cpp"),
                           *srcs, embed, "-o", exe])
    lines = []
    for seed in range(300):
        lines.append(prefix(R.make(seed)[0]))
        lines.append(prefix(R.make_typed(seed)[0]))
        if seed < 150:
            lines.append(prefix(R.make_cmp(seed + 6000)[0]))          # comparison / logical operators among the arithmetic (round 6)
    for g in (G.df1_cascade(6), G.df1_cascade(7), G.df1_cascade(8), G.df1_cascade(12), G.df1_cascade(16), G.par4_sum(), G.par4_sum_fanout(), G.osc_chain(6), G.cross_wire(),
              G.one_pole_readme(), G.mixed_precision_biquad(), G.complex_mix(), G.df1t(), G.df2t(),
              ("seq", ("in", 1), ("del", 1, 300)), ("seq", ("in", 1), ("add", ("del", 1, 40), ("del", 1, 5000))),
              ("seq", ("in", 1), ("add", ("del", 1, 12), ("del", 1, 700))),                       # far line + mid-range read
              G.seq(G.df1_cascade(6), G.mul(G.lit(0.7), G.IN(1))),                               # stage packing with a suffix
              G.seq(G.osc_chain(6), G.add(G.mul(G.lit(0.6), G.IN(1)), G.mul(G.lit(0.3), G.DEL(1, 2)))),
              G.seq(G.df1_cascade(4), G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.mul(G.lit(0.5), G.IN(2))))),
              G.seq(G.df1_cascade(4), G.sub(G.IN(1), G.mul(G.lit(0.25), G.DEL(1, 5)))),
              G.hard_clipper(), G.clipped_biquad(), G.seq(G.clipped_biquad(), G.df1_cascade(2)),
              # the graphs the shipped reference misroutes (test/tests.cpp:67-77) and nested feedbacks around them: feedback_promise_inputs
              G.fb(G.seq(G.add(G.IN(1), G.DEL(2, 1)), G.add(G.DEL(1, 1), G.IN(2)))), G.fb(G.seq(G.add(G.IN(1), G.DEL(3, 1)), G.add(G.DEL(1, 1), G.IN(2)))),
              G.seq(G.fb(G.seq(G.add(G.IN(1), G.DEL(2, 1)), G.add(G.DEL(1, 1), G.IN(2)))), G.fb(G.add(G.mul(G.lit(0.5), G.DEL(1, 1)), G.IN(2))))):
        lines.append(prefix(g))
    # malformed: delay-free loop, missing wires, complex into delay, complex with double, bad arity operands
    for g in (("fb", ("add", ("in", 1), ("in", 2))), ("seq", ("in", 1), ("in", 3)), ("fb", ("in", 1)),
              ("seq", ("mul", ("litc", 1.0, 0.0), ("in", 1)), ("del", 1, 1)), ("mul", ("litc", 1.0, 0.0), ("lit64", 2.0)),
              ("add", ("chan", ("in", 1), ("in", 1)), ("in", 1)), ("del", 1, 0), ("in", 0), ("bogus",)):
        lines.append(prefix(g))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr[-4000:]
    assert "lowered" in out.stdout
    n_low = int(out.stdout.split("lowered")[1].split()[0])
    assert n_low > 640, out.stdout
    assert int(out.stdout.split("wave-split bodies")[1].split()[0]) >= 15, out.stdout   # role extraction + the bodies of every split
