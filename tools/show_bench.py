import json, sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"])
for k,v in d.items():
    if isinstance(v,dict) and "best_plan" in v:
        bp=v[v["best_plan"]]
        print(k, bp.get("kernel"), bp.get("avg_launch_ms"), bp.get("frac"), v.get("parity"), {p:v[p]["frac"] for p in ("library_default","tuned") if p in v}, bp.get("traffic"))
print(d["cpu_baseline"])
