"""Side-by-side of two bench.py JSON lines (profiles/r01_bench_v*.json): every timed row, ratio new/old."""
import json, sys
a, b = (json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:3])
rows = [("headline ms_per_step", a["ms_per_step"], b["ms_per_step"])]
for k, v in b.get("extra", {}).items():
    if isinstance(v, dict):
        for kk in ("ms", "ms_hip_graph", "ms_eager"):
            if kk in v and k in a.get("extra", {}) and kk in a["extra"][k]:
                rows.append((f"{k}.{kk}", a["extra"][k][kk], v[kk]))
    elif k in a.get("extra", {}):
        rows.append((k, a["extra"][k], v))
for name, x, y in rows:
    print(f"{name:44s} {x:14.6g} {y:14.6g}  x{y / x:.3f}")
