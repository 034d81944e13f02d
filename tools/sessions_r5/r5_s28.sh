#!/bin/bash
# round 5, session 28: instruction mix and wait counters of the two inflate kernels (rocprofv3 --pmc, own passes, no tracing)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O /dev/shm/gb
R=$(pwd)
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 14000000 7 12 91 16 > /dev/null
export TMPDIR=/tmp
: > $O/s28_inflate_pmc.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  d=$O/s28_pmc; rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $set -d $d --output-format csv -- python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam 64000 > $O/s28_last.log 2>&1 || { echo "set [$set] failed" >> $O/s28_inflate_pmc.txt; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O/s28_inflate_pmc.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("bgzf_decode_kernel", "bgzf_resolve_kernel"):
        if k in r["Kernel_Name"]: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c,v in acc[k].items(): print("%-20s %-26s per launch %.5g  (launches %d)" % (k, c, sum(v)/len(v), len(v)))
PY
  rm -rf $d
done
rm -rf /dev/shm/gb
cat $O/s28_inflate_pmc.txt
