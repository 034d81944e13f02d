#!/bin/bash
# diagnostic: end-to-end CLI time against the device's upper share of the BGZF inflate (FASTF_GPU_INFLATE_MAX)
# usage (GPU box): tools/e2e_share.sh <records> <seq_len> <share> [<share> ...]
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; SL=$2; shift 2
mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
if [ "$SL" = skinny ]; then $R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 0 16; else $R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 $SL 16; fi
for rep in 1 2; do for sh in "$@"; do
  t0=$(date +%s.%N)
  env FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_HOST_THREADS=16 FASTF_GPU_INFLATE_MAX=$sh $R/fastf_amd/bin/fastF bam2db -b /dev/shm/gb/in.bam -a /dev/shm/gb/bar.tsv -f /dev/shm/gb/feat.tsv -o /dev/shm/gb/out -c 0.5 -r 0.5 2> /dev/shm/gb/err.txt >/dev/null
  t1=$(date +%s.%N)
  echo "share_max $sh: wall $(python3 -c "print(round($t1-$t0,3))") s | $(grep -o 'total so far [0-9.]* s' /dev/shm/gb/err.txt | tail -1) | $(grep -o 'BAM decode+pack [0-9.]* s' /dev/shm/gb/err.txt) | $(grep -o "final device share [0-9.]*" /dev/shm/gb/err.txt | head -1)"
done; done
rm -rf /dev/shm/gb
