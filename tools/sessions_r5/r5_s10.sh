#!/bin/bash
# round 5, session 10: resolve without waits between a match's steps; kernel times; windows; the e2e legs of bench.py
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_records.py -x -q > $O/s10_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s10_tests.txt && rc=99
tail -2 $O/s10_tests.txt
[ $rc -ne 0 ] && exit $rc
R=$(pwd); mkdir -p /dev/shm/gb
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 14000000 7 12 91 16 > /dev/null
export TMPDIR=/tmp
for nb in 64000; do
  echo "== $nb blocks"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s10_prof --output-format csv -- python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam $nb > $O/s10_prof_$nb.log 2>&1
  f=$(find $O/s10_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/s10_inflate_kernel_stats_$nb.csv && python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bgzf' in r['Name']: print('   %-22s calls %s avg %.2f ms min %.2f max %.2f' % (r['Name'][:22], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))"
  rm -rf $O/s10_prof
done
rm -rf /dev/shm/gb
timeout -k 10 900 python3 bench.py --no-cpu --no-devpath --steps 5 --warmup 1 > $O/s10_bench.json 2> $O/s10_bench.err || { tail -5 $O/s10_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/s10_bench.json').read().strip().splitlines()[-1])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for var in ('host_inflate','hybrid_inflate'):
            x=v.get(var,{})
            print(k,var,"%.2f s, %.1f M rec/s to exit; steady %.1f M rec/s; start-up %.2f s; md5 %s" % (x.get('seconds',0), x.get('value',0)/1e6, (x.get('steady_state_records_per_s') or 0)/1e6, x.get('start_up_s') or 0, x.get('matrix_md5')))
            if var=='hybrid_inflate': print("   ", x.get('reader','')[:500]); print("   ", x.get('stages','')[:400])
PY
