#!/bin/bash
# diagnostic: instruction mix and wait counters of one hot-path kernel on the configs[2] bench steps
# usage (GPU box, repo root): tools/kernel_pmc.sh <kernel-name-substring>      (rocprofv3 PMC passes, no tracing)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
K=${1:-filter_pack_stream}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TA_TCP_STATE_READ TCC_EA0_RDREQ"; do
  d=$R/gpurun_out/pmc_k_$(echo $set | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --pmc $set -d $d --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-devpath > /dev/null 2>&1 || { echo "set [$set] failed"; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$K" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("%-28s per launch %.5g  (launches %d)" % (k, sum(v)/len(v), len(v)))
PY
done
