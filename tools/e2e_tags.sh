#!/bin/bash
# diagnostic: wall time of `fastF crb` and `fastF extract` on a generated Cell-Ranger-shaped BAM:  tools/e2e_tags.sh [records] [seq_len]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-20000000}; SL=${2:-91}
mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 $SL 16 > /dev/null
cd /dev/shm/gb/out
python3 - "$R" $N <<'PY'
import os, subprocess, sys, time
R, N = sys.argv[1], int(sys.argv[2])
env = dict(os.environ, FASTF_HOST_THREADS="16", FASTF_PROFILE="1")
for name, args in (("crb", ["crb", "-b", "/dev/shm/gb/in.bam", "-o", "/dev/shm/gb/out/cb_cr.tsv.gz"]),
                   ("extract UB", ["extract", "-b", "/dev/shm/gb/in.bam", "-t", "UB", "-T", "0"]),
                   ("extract xf (integer)", ["extract", "-b", "/dev/shm/gb/in.bam", "-t", "xf", "-T", "1"])):
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([R + "/fastf_amd/bin/fastF"] + args, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        print("%-22s rc %d  %.3f s  %.1f M records/s  %s" % (name, p.returncode, dt, N / dt / 1e6, " | ".join(l for l in p.stderr.splitlines() if l.startswith("[tags]"))[:400]), flush=True)
PY
rm -rf /dev/shm/gb
