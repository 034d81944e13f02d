#!/bin/bash
# round 6, session 12: the wave-per-sub-stream generator that leaves decisions (mt_decide_multi_kernel) — kernel tests, the full-size
# e2e tests (push path: ring with wrap), then the step with its draw generation inside the clock
set -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider > $O/s12_tests.txt 2>&1; rc=$?
grep -E "gpu unit|passed|failed" $O/s12_tests.txt | tail -16
[ $rc -ne 0 ] && { grep -v "^\[gpu unit\]" $O/s12_tests.txt | tail -60; exit $rc; }
python3 bench.py --no-cpu --no-e2e --no-devpath --steps 20 --warmup 3 > $O/s12_bench.json 2> $O/s12_bench.err || { tail -5 $O/s12_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/s12_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
print(json.dumps(d.get("step_with_draw_generation"))[:400])
PY
