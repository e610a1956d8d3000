#!/usr/bin/env python3
"""Round 6 (GPU box): do the kernels that cannot take the row walk in lockstep keep their rate from one allocation to the next?
LDS-ring combs, the oscillator chain on plain rows, 64-sample windows with new coefficients per window -- each on SIX fresh allocations
(other allocations of odd sizes in between), the library default next to lockstep geometries that fit their LDS / registers.
   usage: exp_r06_placement.py [ldsring|osc|blocks64|cascade2] ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from zignal_amd import flowz as F, workloads as W  # noqa: E402

L, G, SP, P3 = F.C.FZ_VF_LOCKSTEP, F.C.FZ_VF_GRID_SYNC, F.C.FZ_VF_STAGE_PACK, F.C.FZ_VF_PREFETCH3
ns, T = 1 << 20, 4096
CASES = {
    "ldsring": (W.lds_ring_comb(), {"default": None, "lock256 u32": (1, 32, 256, L | G), "lock256 u16": (1, 16, 256, L | G), "lock128 u32": (1, 32, 128, L | G),
                                    "wg-lockstep256 u32": (1, 32, 256, L), "free u16": (1, 16, 256, 0)}),
    "osc": (W.osc_chain(6), {"default": None, "free p2u8": (2, 8, 256, 0), "free p1u16 packed": (1, 16, 256, SP), "lock p1 packed": (1, 4, 1024, L | G | SP),
                             "lock p2u1": (2, 1, 1024, L | G | P3)}),
    "blocks64": (W.df1_cascade_params(6), {"default": None, "lock p1 packed": (1, 4, 1024, L | G | SP), "lock p2u2": (2, 2, 1024, L | G), "free p1u16 packed": (1, 16, 256, SP)}),
}
which = sys.argv[1:] or list(CASES)
keep = []
for name in which:
    graph, variants = CASES[name]
    prog = F.compile(F.from_sexpr(graph))
    blocks = name == "blocks64"
    Lw = 64
    for trial in range(6):
        if trial:
            keep.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))      # shifts what the next allocations get
        x = torch.empty((T, ns, 1), dtype=torch.float32, device="cuda")
        y = torch.empty((T, ns, prog.n_out), dtype=torch.float32, device="cuda")
        if name == "osc":
            x.zero_(); x[0].fill_(1.0)
        else:
            F.synth_fill(x, 20160512)
        st = torch.zeros((max(prog.n_state, 1), ns), dtype=torch.float32, device="cuda")
        pd = pb = bank = None
        if prog.n_param and not blocks:
            pd = torch.from_numpy(W.osc_chain_params(20160513, np.arange(ns))).cuda()
        if blocks:
            nb = T // Lw
            one = torch.from_numpy(np.ascontiguousarray(W.osc_chain_params(20160520, np.arange(ns))[1:])).cuda()
            pb = one.unsqueeze(0).expand(nb, -1, -1).contiguous()
            bank = prog.bank(ns)
        b_alg = ns * (4 * T * 2 + ((T // Lw) * (8 * prog.n_state + 4 * prog.n_param) if blocks else 8 * prog.n_state + 4 * prog.n_param))
        row = {"graph": name, "trial": trial, "x-y mod 16MiB (MiB)": ((x.data_ptr() - y.data_ptr()) % (16 << 20)) / (1 << 20)}
        for vn, v in variants.items():
            vv = F.make_variant(*v) if v else None

            def run(xx=x, yy=y):
                if blocks:
                    bank.process_blocks(xx, yy, Lw, pb, variant=vv)
                else:
                    prog.run_block(xx, state=st, params=pd, out=yy, variant=vv)
            try:
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record()
                torch.cuda.synchronize()
                row[vn] = round(b_alg / (e0.elapsed_time(e1) / 5) / 1e6 / 8000, 4)
                if trial == 0 and not blocks:
                    row[vn + " kernel"] = prog.kernel_name(vv, ns, T).replace("fz_block_kernel_", "")
            except F.FlowzError as e:
                row[vn] = "refused: " + str(e)[:60]
        if name == "ldsring":                                # in place (n_in == n_out): reads and writes share their pages
            try:
                prog.run_block(x, state=st, out=x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    prog.run_block(x, state=st, out=x)
                e1.record()
                torch.cuda.synchronize()
                row["default in place"] = round(b_alg / (e0.elapsed_time(e1) / 5) / 1e6 / 8000, 4)
            except F.FlowzError as e:
                row["default in place"] = "refused: " + str(e)[:60]
        print(json.dumps(row), flush=True)
        del x, y, st, pd, pb, bank
        torch.cuda.empty_cache()
