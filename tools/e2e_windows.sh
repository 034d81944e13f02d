#!/bin/bash
# diagnostic: window lines of the CLI on a generated Cell-Ranger-shaped BAM for several settings:  tools/e2e_windows.sh <records> <seq_len> "VAR=x VAR=y" "VAR=z" ...
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-20000000}; SL=${2:-91}
shift 2 || true
mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 $SL 16 > /dev/null
for s in "$@"; do
  echo "== $s"
  for rep in 1 2; do
    rm -f /dev/shm/gb/out/*
    env FASTF_PROFILE=1 FASTF_BAM_PROFILE=2 FASTF_HOST_THREADS=16 $s $R/fastf_amd/bin/fastF bam2db -b /dev/shm/gb/in.bam -a /dev/shm/gb/bar.tsv -f /dev/shm/gb/feat.tsv -o /dev/shm/gb/out -c 0.5 -r 0.5 2>&1 >/dev/null | python3 -c "
import re, sys
w = []
for l in sys.stdin:
    m = re.match(r'\[bam\] ([0-9.]+) window (\d+): (\d+) blocks, (\d+) on the device in ([0-9.]+) ms(?: \(queued a window ago\))?, (\d+) on the host in ([0-9.]+) ms', l)
    if m: w.append((float(m.group(1)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(7))))
    if l.startswith('[bam2db] lists'): tot = l.split('total so far')[1].strip(); stages = l.strip()
    if l.startswith('[bam2db] phases'): phases = l.strip()
    if 'device side ready' in l: ready = l.split()[1]
if len(w) > 3:
    full = w[1:-1]                                  # between the first shared window and the last (partial) one
    blocks = sum(x[1] for x in full); span = full[-1][0] - w[0][0]
    print('   device ready %s, %d shared windows, steady state %d blocks in %.1f ms = %.0f blocks/ms (device %.0f%% of them, %.1f ms per window; host %.1f ms), total %s' % (
          ready, len(w), blocks, span * 1e3, blocks / (span * 1e3), 100.0 * sum(x[2] for x in full) / blocks, sum(x[3] for x in full) / len(full), sum(x[4] for x in full) / len(full), tot))
    print('      ' + stages); print('      ' + phases)
"
  done
done
rm -rf /dev/shm/gb
