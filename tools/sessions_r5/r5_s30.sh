#!/bin/bash
# round 5, session 30: kernel times (rocprofv3 --kernel-trace --stats) of the lane decoder at 64 / 32 / 16 blocks per wave and with half
# the blocks in flight per CU (twice the LDS asked for): what is the decode bound by?
set -o pipefail
O=gpurun_out/r5; mkdir -p $O /dev/shm/gb
R=$(pwd)
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 28000000 7 12 91 16 > /dev/null
export TMPDIR=/tmp
: > $O/s30_lpw_kernels.txt
for v in main gi2_lpw32 gi2_lpw32_half gi2_lpw16 gi2_lpw16_half main; do
  for nb in 64000 128000; do
    echo "== $v, $nb blocks" >> $O/s30_lpw_kernels.txt
    [ $v = main ] && unset FASTF_LIB_OVERRIDE || export FASTF_LIB_OVERRIDE=$R/build/$v/libfastf_amd.so
    rm -rf $O/s30_prof
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s30_prof --output-format csv -- python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam $nb > $O/s30_last.log 2>&1
    f=$(find $O/s30_prof -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'bgzf' in r['Name'] or 'gr_crc' in r['Name']: print('   %-22s calls %s avg %.2f ms min %.2f max %.2f' % (r['Name'][:22], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))" >> $O/s30_lpw_kernels.txt
  done
done
rm -rf /dev/shm/gb $O/s30_prof
cat $O/s30_lpw_kernels.txt
