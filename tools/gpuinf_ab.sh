#!/bin/bash
# diagnostic: A/B of device-inflate builds on one generated Cell-Ranger-shaped BAM: tools/gpuinf_ab.sh <records> <blocks> <variant> [<variant> ...]
# (variants are directories under build/ made by tools/build_variant.sh; "main" is the in-tree library)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-6000000}; B=${2:-16000}; shift 2 || true
mkdir -p /dev/shm/gb
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 91 16 > /dev/null
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then python3 $R/tools/gpuinf_bench.py /dev/shm/gb/cr.bam $B 2>&1 | tail -3
  else FASTF_LIB_OVERRIDE=$R/build/$v/libfastf_amd.so python3 $R/tools/gpuinf_bench.py /dev/shm/gb/cr.bam $B 2>&1 | tail -3; fi
done
rm -rf /dev/shm/gb
