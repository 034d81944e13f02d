#!/bin/bash
# round 5, session 13: the parallel MT19937 (jump-ahead): parity with the host stream, the bench line with draw generation inside the clock
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/s13_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s13_tests.txt && rc=99
tail -4 $O/s13_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py --no-e2e --no-cpu --steps 30 --warmup 3 > $O/s13_bench.json 2> $O/s13_bench.err || { tail -5 $O/s13_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/s13_bench.json').read().strip().splitlines()[-1])
print("step %.3f ms; read_frac %.3f; devpath %.3f G rec/s" % (d['ms_per_step'], d['whole_path']['read_frac_of_peak'], (d.get('device_path') or {}).get('value',0)/1e9))
print(json.dumps(d['step_with_draw_generation'], indent=1))
PY
