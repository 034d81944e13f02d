"""one-line view of a bench.py JSON line (stdin): value, ms/step, per-kernel ms"""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"] / 1e9, 3), "Grec/s", round(d["ms_per_step"], 4), "ms  frac", round(d["roofline"]["frac"], 3),
          {k: round(v, 4) for k, v in d["kernels_ms"].items()})
