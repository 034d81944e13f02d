#!/bin/bash
# round 5, session 14: the jump kernel with sixteen waves on the convolution
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "mt or decision" > $O/s14_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s14_tests.txt && rc=99
tail -2 $O/s14_tests.txt
[ $rc -ne 0 ] && exit $rc
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/s14_prof --output-format csv -- python3 bench.py --no-e2e --no-cpu --no-devpath --steps 10 --warmup 2 > $O/s14_bench.json 2> $O/s14_bench.err || { tail -5 $O/s14_bench.err; exit 1; }
f=$(find $O/s14_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'mt_' in r['Name'] or 'draw_bits' in r['Name']: print('   %-30s calls %s avg %.3f ms' % (r['Name'][:30], r['Calls'], float(r['AverageNs'])/1e6))"
rm -rf $O/s14_prof
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/s14_bench.json') if l.startswith('{"metric')][-1])
w=d['step_with_draw_generation']; print("step %.3f ms; with draws %.3f ms; generation %.3f ms = %.1f G draws/s; same %s" % (d['ms_per_step'], w['ms_per_step'], w['draw_generation_ms'], w['draws_per_s']/1e9, w['same_counters_as_the_resident_stream']))
PY
