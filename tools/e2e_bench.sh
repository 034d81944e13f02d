#!/bin/bash
# End-to-end CLI throughput (BAM file -> three .gz outputs) on the GPU box.
# usage: tools/e2e_bench.sh [n_records] [seq_len]     seq_len > 0: records shaped like Cell Ranger's (bases, qualities, 14 tags)
set -e
N=${1:-10000000}
SL=${2:-0}
R=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/fastf_e2e; mkdir -p $W/out
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from fastf_amd import synth
bt, ft, _, _ = synth.make_lists(10000, 30000, seed=4242)
open("$W/bar.tsv", "wb").write(bt); open("$W/feat.tsv", "wb").write(ft)
PY
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam $W/in.bam $W/bar.tsv $W/feat.tsv $N 7 10 $SL 16
ls -la $W/in.bam | awk '{print "BAM bytes", $5}'
echo "host cores visible: $(nproc)"
run() {
  rm -f $W/out/*.gz
  local t0=$(date +%s.%N)
  env "$@" $R/fastf_amd/bin/fastF bam2db -b $W/in.bam -a $W/bar.tsv -f $W/feat.tsv -o $W/out -c 1 -r 1 > $W/log.txt 2> $W/err.txt || { cat $W/err.txt; return 1; }
  local t1=$(date +%s.%N)
  python3 -c "dt=$t1-$t0; print('  E2E %-60s %.3f s  %.2f M records/s' % ('$*', dt, $N/dt/1e6))"
  grep -E "^\[bam" $W/err.txt | sed 's/^/    /'
}
run FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_HOST_THREADS=1
run FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_INFLATE=zlib
run FASTF_PROFILE=1 FASTF_BAM_PROFILE=1
run FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_GZIP_LEVEL=1
run FASTF_PROFILE=1 FASTF_BAM_PROFILE=1 FASTF_GZIP_LEVEL=1 FASTF_HOST_THREADS=32
grep -E "fastQ reads" $W/log.txt | sed 's/^/    /'
zcat $W/out/matrix.mtx.gz | md5sum
