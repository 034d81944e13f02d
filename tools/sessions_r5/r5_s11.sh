#!/bin/bash
# round 5, session 11: where a token-step of the lane decoder spends its cycles (s_memtime stamps, experiment build)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O /dev/shm/gb
R=$(pwd)
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 6000000 7 12 91 16 > /dev/null
FASTF_LIB_OVERRIDE=$R/build/gi2_stamps/libfastf_amd.so timeout -k 10 200 python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam 16000 > $O/s11_stamps.txt 2>&1
tail -4 $O/s11_stamps.txt
rm -rf /dev/shm/gb
