#!/bin/bash
# round 5, session 29: blocks per wave of the lane decoder, 64 (four workgroups per CU: one wave per SIMD) against 48 (five) and 32
# (eight: two waves per SIMD, each with half its lanes) — the counters say the wave waits 37 % of its cycles and issues an
# instruction every seven
set -o pipefail
O=gpurun_out/r5; mkdir -p $O /dev/shm/gb
R=$(pwd)
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 28000000 7 12 91 16 > /dev/null
: > $O/s29c_lpw.txt
for v in main gi2_lpw32 gi2_lpw32_half gi2_lpw16 gi2_lpw16_half; do
  for nb in 45000 128000; do
    echo "== $v, $nb blocks" >> $O/s29c_lpw.txt
    if [ $v = main ]; then timeout -k 10 200 python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam $nb 2>&1 | grep "^run" >> $O/s29c_lpw.txt
    else FASTF_LIB_OVERRIDE=$R/build/$v/libfastf_amd.so timeout -k 10 200 python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam $nb 2>&1 | grep "^run" >> $O/s29c_lpw.txt; fi
  done
done
rm -rf /dev/shm/gb
cat $O/s29c_lpw.txt
