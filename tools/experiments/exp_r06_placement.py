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
    "ldsring": (W.lds_ring_comb(), {"default": None, "free u32": (1, 32, 256, 0), "lock256 u32": (1, 32, 256, L | G), "lock256 u16": (1, 16, 256, L | G), "lock256 u8": (1, 8, 256, L | G),
                                    "lock256 u24": (1, 24, 256, L | G)}),
    "osc": (W.osc_chain(6), {"default": None, "free p2u8": (2, 8, 256, 0), "free p1u16 packed": (1, 16, 256, SP), "lock p1 packed": (1, 4, 1024, L | G | SP),
                             "lock p2u1": (2, 1, 1024, L | G | P3), "free p2u16": (2, 16, 256, 0)}),
    "blocks64": (W.df1_cascade_params(6), {"default": None, "free p1u16 packed": (1, 16, 256, SP), "free p2u8": (2, 8, 256, 0)}),
}
which = sys.argv[1:] or list(CASES)
keep = []


def med(v):
    return sorted(v)[len(v) // 2]


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

        def runner(v, inplace=False):
            vv = F.make_variant(*v) if v else None
            if blocks:
                return lambda: bank.process_blocks(x, y, Lw, pb, variant=vv)
            return lambda: prog.run_block(x, state=st, params=pd, out=x if inplace else y, variant=vv)
        runs = {}
        for vn, v in variants.items():
            try:
                runs[vn] = runner(v)
                runs[vn]()
                if trial == 0 and not blocks:
                    row[vn + " kernel"] = prog.kernel_name(F.make_variant(*v) if v else None, ns, T).replace("fz_block_kernel_", "")
            except F.FlowzError as e:
                row[vn] = "refused: " + str(e)[:60]
                runs.pop(vn, None)
        if name == "ldsring":                                # in place (n_in == n_out): reads and writes share their pages
            runs["default in place"] = runner(None, True)
        torch.cuda.synchronize()
        # (the first batch of round 6 timed each variant once, right after the allocation: the variant that came first ran ~7 % slower than the same
        #  kernel a second later -- clocks and queues still settling.  Now: >= 300 ms of launches first, then three interleaved rounds, the median)
        t_end = torch.cuda.Event(enable_timing=True); t0 = torch.cuda.Event(enable_timing=True)
        import time
        tw = time.time()
        while time.time() - tw < 0.3:
            runs["default"]()
            torch.cuda.synchronize()
        times = {vn: [] for vn in runs}
        for _ in range(3):
            for vn, fn in runs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[vn].append(e0.elapsed_time(e1) / 5)
        for vn in runs:
            row[vn] = round(b_alg / med(times[vn]) / 1e6 / 8000, 4)
        print(json.dumps(row), flush=True)
        del x, y, st, pd, pb, bank, runs
        torch.cuda.empty_cache()
