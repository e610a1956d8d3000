"""TESTS ONLY: evaluate the product's lowered IR (fz_program_ir) with numpy.

Lets the GPU-less container check the lowering (wiring, feedback resolution, sharing, state
layout) against the oracle.  It is not part of the product: zignal_amd never interprets IR on
the CPU."""
import numpy as np

F32 = np.float32


def _join(lo, hi):
    """two float32 words -> float64 (typed programs carry doubles as (low word, high word))"""
    w = np.empty(lo.shape + (2,), F32)
    w[..., 0], w[..., 1] = lo, hi
    return w.view(np.float64)[..., 0]


def _split(d):
    w = np.ascontiguousarray(d, np.float64)[..., None].view(F32)
    return w[..., 0].copy(), w[..., 1].copy()


def run_ir(prog, x, params=None, state=None, mod=None):
    """x: [T, ns, n_in] float32 slots -> (y [T, ns, n_out] float32 slots, state [n_state, ns]) following the documented
    state layout: lines in fz_program_lines order, a float line takes `depth` rows (row+j = value at t-1-j), a double
    line (typed programs) 2*depth rows: slot j is ONE row of ns doubles = float rows (row+2j, row+2j+1)."""
    x = np.asarray(x, F32)
    if x.ndim == 2:
        x = x[:, :, None]
    T, ns, _ = x.shape
    ir = prog.ir()
    dts = prog.ir_dtypes()
    outs = prog.outputs()
    codes = prog.output_slot_codes()
    lines = prog.lines()
    ldt = prog.line_dtypes()
    row0, r = {}, 0
    for (src, depth), dt in zip(lines, ldt):
        f64 = dt in ("f64", "re64", "im64")
        row0[src] = (r, depth, f64)
        r += depth * (2 if f64 else 1)
    if state is None:
        state = np.zeros((max(r, 1), ns), F32)
    else:
        state = np.array(state, F32, copy=True)
    flat = state.reshape(-1)

    def drow(rr):                                   # the row of ns doubles that starts at float row rr
        return flat[rr * ns:(rr + 2) * ns].view(np.float64)

    y = np.empty((T, ns, len(outs)), F32)
    with np.errstate(all="ignore"):
        for t in range(T):
            v = [None] * len(ir)
            for i, (kind, a, b, val) in enumerate(ir):
                if kind == "input":
                    v[i] = _join(x[t, :, a], x[t, :, a + 1]) if dts[i] == "f64" else x[t, :, a]
                elif kind == "const":
                    v[i] = np.full(ns, val, np.float64) if dts[i] == "f64" else np.full(ns, F32(val), F32)
                elif kind == "param":
                    v[i] = np.asarray(params[a], F32)
                elif kind == "mod":
                    v[i] = np.full(ns, F32(mod[a][t]), F32)
                elif kind == "delay":
                    r0, depth, f64 = row0[a]
                    assert 1 <= b <= depth
                    v[i] = drow(r0 + 2 * (b - 1)).copy() if f64 else state[r0 + b - 1].copy()
                    assert (dts[i] == "f64") == f64
                elif kind == "add":
                    v[i] = v[a] + v[b]
                elif kind == "sub":
                    v[i] = v[a] - v[b]
                elif kind == "mul":
                    v[i] = v[a] * v[b]
                elif kind == "div":
                    v[i] = v[a] / v[b]
                elif kind == "neg":
                    v[i] = -v[a]
                elif kind in ("lt", "le", "gt", "ge", "eq", "ne"):      # a float 1.0 / 0.0; operands compared in their common type
                    m = {"lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal, "eq": np.equal, "ne": np.not_equal}[kind](v[a], v[b])
                    v[i] = np.where(m, F32(1), F32(0)).astype(F32)
                elif kind == "abslt":
                    v[i] = (np.abs(v[a]) < np.abs(v[b])).astype(v[a].dtype)
                elif kind == "select":
                    v[i] = np.where(v[a] != 0, v[b], v[val])
                elif kind == "widen":
                    v[i] = np.asarray(v[a], F32).astype(np.float64)
                elif kind == "narrow":
                    v[i] = np.asarray(v[a], np.float64).astype(F32)
                else:
                    raise AssertionError(kind)
                assert np.asarray(v[i]).dtype == (np.float64 if dts[i] == "f64" else F32), (i, kind, dts[i])
            for j, o in enumerate(outs):
                if codes[j] in (4, 6, 8):
                    y[t, :, j] = _split(v[o])[0]
                elif codes[j] in (5, 7, 9):
                    y[t, :, j] = _split(v[o])[1]
                else:
                    y[t, :, j] = v[o]                       # (a double narrows to the float frame, code 1)
            for src, (r0, depth, f64) in row0.items():
                if f64:
                    for j in range(depth - 1, 0, -1):
                        drow(r0 + 2 * j)[:] = drow(r0 + 2 * (j - 1))
                    drow(r0)[:] = v[src]
                else:
                    state[r0 + 1:r0 + depth] = state[r0:r0 + depth - 1].copy()
                    state[r0] = v[src]
    return y, state
