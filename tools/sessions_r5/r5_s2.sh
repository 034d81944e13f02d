#!/bin/bash
# round 5, session 2: the two-kernel device inflate — parity tests, kernel rate against round 4's wave kernel, the e2e legs
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_records.py tests/test_gpu_e2e.py -x -q > $O/s2_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s2_tests.txt && rc=99
tail -5 $O/s2_tests.txt
[ $rc -ne 0 ] && exit $rc
R=$(pwd); mkdir -p /dev/shm/gb
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread
build/gen_bam /dev/shm/gb/cr.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 6000000 7 12 91 16 > /dev/null
{ echo "== two kernels (lanes)"; timeout -k 10 120 python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam 16000 2>&1 | tail -3
  echo "== FASTF_GI_KERNEL=wave"; FASTF_GI_KERNEL=wave timeout -k 10 120 python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam 16000 2>&1 | tail -3; } > $O/s2_inflate_ab.txt 2>&1
cat $O/s2_inflate_ab.txt
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s2_prof --output-format csv -- python3 tools/gpuinf_bench.py /dev/shm/gb/cr.bam 16000 > $O/s2_prof.log 2>&1
f=$(find $O/s2_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/s2_inflate_kernel_stats.csv && head -8 $f
rm -rf /dev/shm/gb $O/s2_prof
timeout -k 10 500 python3 bench.py --no-cpu --no-devpath --steps 5 --warmup 1 > $O/s2_bench.json 2> $O/s2_bench.err || { tail -5 $O/s2_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/s2_bench.json').read().strip().splitlines()[-1])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for var in ('host_inflate','hybrid_inflate'):
            x=v.get(var,{})
            print(k,var,"%.2f s, %.1f M rec/s to exit; steady %.1f M rec/s; md5 %s" % (x.get('seconds',0), x.get('value',0)/1e6, (x.get('steady_state_records_per_s') or 0)/1e6, x.get('matrix_md5')))
            print("   ", x.get('reader','')[:400])
PY
