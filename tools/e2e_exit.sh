#!/bin/bash
# diagnostic: the CLI on a generated Cell-Ranger-shaped BAM, wall time to process exit with the stage and close lines
#   tools/e2e_exit.sh [records] [seq_len] [VAR=x ...]      (each extra argument is an environment assignment; several runs: separate the sets with ---)
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
sets=(); cur=""
for a in "$@"; do if [ "$a" = "---" ]; then sets+=("$cur"); cur=""; else cur="$cur $a"; fi; done
sets+=("$cur")
for rep in 1 2; do for s in "${sets[@]}"; do
  rm -f /dev/shm/gb/out/*
  python3 - "$R" $s <<'PY'
import os, subprocess, sys, time
R = sys.argv[1]; env = dict(os.environ, FASTF_PROFILE="1", FASTF_BAM_PROFILE="1", FASTF_HOST_THREADS="16")
for a in sys.argv[2:]:
    k, v = a.split("=", 1); env[k] = v
t0 = time.perf_counter(); w0 = time.time()
p = subprocess.run([R + "/fastf_amd/bin/fastF", "bam2db", "-b", "/dev/shm/gb/in.bam", "-a", "/dev/shm/gb/bar.tsv", "-f", "/dev/shm/gb/feat.tsv",
                    "-o", "/dev/shm/gb/out", "-c", "0.5", "-r", "0.5"], env=env, capture_output=True, text=True)
wall = time.perf_counter() - t0
closed = [float(l.split(" at ")[1].split()[0]) - w0 for l in p.stderr.splitlines() if "outputs closed at" in l]
print("== %s: exit %.3f s, outputs closed %.3f s, rc %d" % (" ".join(sys.argv[2:]) or "(default)", wall, closed[-1] if closed else -1, p.returncode))
for l in p.stderr.splitlines():
    if l.startswith("[bam2db] lists") or l.startswith("[bam]") or "teardown" in l: print("   ", l[:460])
PY
done; done
rm -rf /dev/shm/gb
