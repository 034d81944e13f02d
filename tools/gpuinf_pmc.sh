#!/bin/bash
# diagnostic: instruction mix of bgzf_inflate_kernel (rocprofv3 PMC passes, no tracing)
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
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  d=$R/gpurun_out/pmc_gi_$(echo $set | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --pmc $set -d $d --output-format csv -- python3 $R/tools/gpuinf_bench.py /dev/shm/gb/cr.bam 8000 > /dev/null 2>&1 || true
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "bgzf_inflate" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("%-24s per launch %.4g  (launches %d)" % (k, sum(v)/len(v), len(v)))
PY
done
rm -rf /dev/shm/gb
