#!/usr/bin/env python3
"""Print every object of a bench line that carries a roofline fraction: path, kernel, ms, frac, parity."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
if len(d) == 1 and isinstance(next(iter(d.values())), dict):
    d = next(iter(d.values()))
r = d.get("roofline")
if r:
    print(f"headline value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  frac {r['frac']}  kernel {r['kernel']}  row-walk {r.get('measured_row_walk_GBs')} "
          f"({r.get('frac_of_row_walk')})  copy {r.get('measured_copy_GBs')}  limiter {r.get('limiter')}  parity {d.get('parity')}")
    s = r.get("sustained")
    if s:
        print(f"  sustained {s['avg_launch_ms']} ms frac {s['frac']} board {s.get('board')} J/launch {s.get('joules_per_launch')}")


def walk(o, path):
    if not isinstance(o, dict):
        return
    plans = [k for k in ("library_default", "tuned", "forced") if k in o]
    if plans:
        txt = "  ".join(f"{k[:7]} {o[k]['avg_launch_ms']:.4f} ms {o[k]['frac']:.4f} {o[k]['kernel'].replace('fz_block_kernel_', '')}"
                        + (f" traffic x{o[k]['traffic'] / o['algorithmic_bytes_per_launch']:.4f}" if o[k].get("traffic") else "") for k in plans)
        extra = ""
        if "sustained" in o:
            extra = f"  J/launch {o['sustained'].get('joules_per_launch')}"
        if "limiter" in o:
            extra += f"  limiter {o['limiter']}"
        if "dirac_201" in o:
            extra += "  dirac: " + o["dirac_201"][:20]
        print(f"{path:55s} {txt}  | {o.get('parity', '')[:34]}{extra}")
    for k, v in o.items():
        if k not in ("library_default", "tuned", "forced", "roofline", "cpu_baseline", "config"):
            walk(v, f"{path}.{k}" if path else k)


walk(d, "")
c = d.get("cpu_baseline")
if c:
    print(f"cpu_baseline {c['value']} {c['unit']} on {c['cores']} cores ({c['kind']})")
