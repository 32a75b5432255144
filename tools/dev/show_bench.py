"""dev: the interesting fields of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_fresh", "avg_kernel_ms", "whole_call_frac", "traffic")})
for k, v in d.get("modes", {}).items():
    print("mode", k, {a: b for a, b in v.items() if a not in ("timing", "binding")})
for k, v in d.get("other_configs", {}).items():
    print("cfg", k, {a: b for a, b in v.items() if a not in ("workload", "unit", "dtype", "steps")})
