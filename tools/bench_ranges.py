#!/usr/bin/env python3
"""The `summary` maps of several bench lines side by side and their ranges (one line per object): what DESIGN 6 quotes.
usage: tools/bench_ranges.py profiles/r05/bench_line_*.json"""
import json
import sys

lines = []
for f in sys.argv[1:]:
    ls = [l for l in open(f) if l.startswith("{")]
    if ls:
        d = json.loads(ls[-1])
        if "summary" in d:
            lines.append((f, d))
keys = []
for _, d in lines:
    for k in d["summary"]:
        if k not in keys:
            keys.append(k)
print(f"{'object':42s} " + " ".join(f"{i:>6d}" for i in range(len(lines))) + "   range")
hv = [d["roofline"]["frac"] for _, d in lines]
print(f"{'HEADLINE (roofline.frac)':42s} " + " ".join(f"{v:6.3f}" for v in hv) + f"   {min(hv):.3f}-{max(hv):.3f}")
sv = [(d["roofline"].get("sustained") or {}).get("frac") or float("nan") for _, d in lines]
print(f"{'  sustained':42s} " + " ".join(f"{v:6.3f}" for v in sv))
for k in keys:
    vs = [d["summary"].get(k) for _, d in lines]
    have = [v for v in vs if v is not None]
    print(f"{k:42s} " + " ".join(f"{v:6.3f}" if v is not None else "     -" for v in vs) + f"   {min(have):.3f}-{max(have):.3f}")
for i, (f, d) in enumerate(lines):
    print(f"# {i}: {f}  value {d['value']} {d['unit']}  {d['roofline']['kernel']}  cpu {d.get('cpu_baseline', {}).get('value')} on {d.get('cpu_baseline', {}).get('cores')} cores")
