#!/bin/bash
# diagnostic: kernel and copy durations of the device inflate microbenchmark
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p /dev/shm/gb
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 3000000 7 12 91 16
d=$R/gpurun_out/trace_gi; rm -rf $d
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $d --output-format csv -- python3 $R/tools/gpuinf_bench.py /dev/shm/gb/cr.bam 8000 > /dev/null 2>&1 || true
for f in $(find $d -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do echo $f; head -6 $f; done
f=$(find $d -name "*memory_copy_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[-14:]: print(r.get("Direction"), r.get("Bytes") or r.get("Size"), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), "ns")
PY
rm -rf /dev/shm/gb
