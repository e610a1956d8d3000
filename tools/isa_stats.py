#!/usr/bin/env python3
"""Dev tool: JIT a graph variant (no GPU needed) and print instruction histogram + resources.

usage: tools/isa_stats.py [graph] [P] [U] [block] [flags]   (graph: cascade6|par4|par4f|osc|df1|ring)
"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graphs as G  # noqa: E402
from zignal_amd import flowz as F  # noqa: E402

GRAPHS = {"cascade6": lambda: G.df1_cascade(6), "par4": G.par4_sum, "par4f": G.par4_sum_fanout,
          "osc": lambda: G.osc_chain(6), "gain": lambda: G.mul(G.lit(0.5), G.IN(1)), "cascade2": lambda: G.df1_cascade(2), "cascade4": lambda: G.df1_cascade(4), "df1": G.df1, "integrator": G.integrator,
          "mod6": lambda: G.df1_cascade_modulated(6), "ldsring": G.lds_ring_comb,
          "ring": lambda: G.seq(G.add(G.IN(1), G.mul(G.lit(0.5), G.DEL(1, 40))), G.fb(G.add(G.mul(G.lit(0.7), G.DEL(1, 23)), G.IN(2))))}


def main():
    a = sys.argv[1:]
    name = a[0] if a else "cascade6"
    P, U, B, FL = (int(a[i]) if len(a) > i else d for i, d in ((1, 2), (2, 8), (3, 256), (4, 0)))
    with tempfile.TemporaryDirectory() as td:
        os.environ["FLOWZ_HIP_CACHE"] = td
        p = F.compile(F.from_sexpr(GRAPHS[name]()))
        p.build(F.make_variant(P, U, B, FL))
        f = glob.glob(td + "/*.hsaco")[0]
        llvm = "/opt/rocm/lib/llvm/bin/"
        dis = subprocess.check_output([llvm + "llvm-objdump", "-d", f], text=True)
        notes = subprocess.check_output([llvm + "llvm-readelf", "--notes", f], text=True)
        if len(a) > 5:
            open(a[5], "w").write(dis)
    hist = {}
    for line in dis.splitlines():
        t = line.split()
        if len(t) > 1 and t[0][0] in "vsgbd" and "_" in t[0]:
            hist[t[0]] = hist.get(t[0], 0) + 1
    print(f"{name} P={P} U={U} block={B} flags={FL}: ops/sample={p.n_ops} state={p.n_state}")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"  {v:6d} {k}")
    for line in notes.splitlines():
        if any(s in line for s in (".vgpr_count", ".sgpr_count", "spill_count", "private_segment_fixed", "group_segment_fixed", ".agpr_count")):
            print(" ", line.strip())


if __name__ == "__main__":
    main()
