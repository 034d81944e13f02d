#!/bin/bash
# round 6, session 10: the two-launch jump seating — kernel tests, then the step with its draw generation inside the clock
set -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider > $O/s10_kernels.txt 2>&1; rc=$?
grep -E "gpu unit|passed|failed" $O/s10_kernels.txt | tail -5
[ $rc -ne 0 ] && { tail -40 $O/s10_kernels.txt; exit $rc; }
python3 bench.py --no-cpu --no-e2e --no-devpath --steps 20 --warmup 3 > $O/s10_bench.json 2> $O/s10_bench.err || { tail -5 $O/s10_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/s10_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
print(json.dumps(d.get("step_with_draw_generation"))[:1500])
for k,v in d.get("kernels_ms",{}).items():
    if "mt" in k or "draw" in k: print(k, v)
PY
