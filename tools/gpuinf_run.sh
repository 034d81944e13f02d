#!/bin/bash
# diagnostic: generate a Cell-Ranger-shaped BAM in /dev/shm and time the device inflate on its first blocks
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-6000000}
mkdir -p /dev/shm/gb
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 91 16
python3 $R/tools/gpuinf_bench.py /dev/shm/gb/cr.bam ${2:-16000} 2>&1 | tail -4
rm -rf /dev/shm/gb
