#!/bin/bash
# round 6, session 13: mt_decide_multi_kernel with its LDS reads batched per sweep — kernel tests, the step with its draws, kernel times
set -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider > $O/s13_tests.txt 2>&1; rc=$?
grep -E "gpu unit|passed|failed" $O/s13_tests.txt | tail -4
[ $rc -ne 0 ] && { grep -v "^\[gpu unit\]" $O/s13_tests.txt | tail -60; exit $rc; }
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/s13_prof --output-format csv -- python3 bench.py --no-cpu --no-e2e --no-devpath --steps 10 --warmup 2 > $O/s13_bench.json 2> $O/s13_bench.err || { tail -5 $O/s13_bench.err; exit 1; }
python3 - <<'PY'
import json,glob,csv
d=json.loads([l for l in open("gpurun_out/r6/s13_bench.json") if l.startswith('{"metric"')][-1])
print("ms_per_step", d["ms_per_step"])
print(json.dumps(d.get("step_with_draw_generation"))[:330])
f=glob.glob("gpurun_out/r6/s13_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("mt_", "ring_clear", "draw_bits")): print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
