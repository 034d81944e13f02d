#!/bin/bash
# round 5, session 33 (as session 8, after the LDS-staged pack kernel, the repair kernel and the larger loan): rocprofv3 kernel trace of the CLI itself (60 M Cell-Ranger-shaped records): which kernels the product runs and for how long
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
R=$(pwd); mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 60000000 7 12 91 16 > /dev/null
export TMPDIR=/tmp
FASTF_FULL_TEARDOWN=1 FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_HOST_THREADS=16 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s33_prof --output-format csv -- fastf_amd/bin/fastF bam2db -b /dev/shm/gb/in.bam -a /dev/shm/gb/bar.tsv -f /dev/shm/gb/feat.tsv -o /dev/shm/gb/out -c 0.5 -r 0.5 > $O/s33_cli.log 2>&1
grep "^\[bam\|^\[bam2db" $O/s33_cli.log | cut -c1-600
f=$(find $O/s33_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/s33_cli_kernel_stats.csv && python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    print('   %-44s calls %5s total %8.2f ms avg %8.3f ms' % (r['Name'][:44], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))"
t=$(find $O/s33_prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && cp $t $O/s33_cli_kernel_trace.csv; rm -rf $O/s33_prof /dev/shm/gb
