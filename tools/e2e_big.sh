#!/bin/bash
# diagnostic: the CLI on the configs[2]-sized Cell-Ranger-shaped BAM (20 M-record body x REP behind one header) with the reader's
# per-window lines; usage: tools/e2e_big.sh [REP=10] [VAR=x ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REP=${1:-10}
mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 20000000 7 12 91 16 $REP
ls -la /dev/shm/gb/in.bam
shift || true
env FASTF_PROFILE=${FASTF_PROFILE:-1} FASTF_BAM_PROFILE=2 FASTF_HOST_THREADS=16 "$@" $R/fastf_amd/bin/fastF bam2db -b /dev/shm/gb/in.bam -a /dev/shm/gb/bar.tsv -f /dev/shm/gb/feat.tsv -o /dev/shm/gb/out -c 0.5 -r 0.5 2>&1 >/dev/null | grep -v "^\[bam\] [0-9.]* window [0-9]*[1-9]:" 
rm -rf /dev/shm/gb
