#!/bin/bash
# round 5, session 27: the symbol loop takes two tokens a pass (a literal and what follows it): tests, kernel times, windows
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_records.py -m gpu -x -q > $O/s27_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s27_tests.txt && rc=99
tail -2 $O/s27_tests.txt
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
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s27_prof --output-format csv -- python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam $nb > $O/s27_prof_$nb.log 2>&1
  tail -3 $O/s27_prof_$nb.log
  f=$(find $O/s27_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/s27_inflate_kernel_stats_$nb.csv && python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bgzf' in r['Name']: print('   %-22s calls %s avg %.2f ms min %.2f max %.2f' % (r['Name'][:22], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))"
  rm -rf $O/s27_prof
done
rm -rf /dev/shm/gb
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_X=1" > $O/s27_windows.txt 2>&1
grep -v "phases\|lists" $O/s27_windows.txt | cut -c1-300
