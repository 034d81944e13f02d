#!/bin/bash
# round 5, session 1: the blocked/streaming push path + region-walk first sort pass under the whole GPU suite, a short bench line,
# and the inflate probe (decode alone / far sources from the ring)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/s1_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s1_tests.txt && rc=99
tail -5 $O/s1_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-e2e --no-cpu --steps 50 --warmup 3 > $O/s1_bench.json 2> $O/s1_bench.err || { tail -5 $O/s1_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/s1_bench.json').read().strip().splitlines()[-1])
print("step %.3f ms; value %.1f G rec/s; read_frac %.3f; devpath %s" % (d['ms_per_step'], d['value']/1e9, d['whole_path']['read_frac_of_peak'], (d.get('device_path') or {}).get('value')))
for k,v in d['kernels'].items(): print("  %-40s %.4f ms x %.2f  frac %.3f" % (k[:40], v['avg_ms'], v['launches_per_step'], v['frac']))
PY
timeout -k 10 400 bash tools/gpuinf_ab.sh 6000000 16000 main gi_nocopy gi_nearonly > $O/s1_inflate_probe.txt 2>&1; tail -12 $O/s1_inflate_probe.txt
