#!/bin/bash
# 12 waves per workgroup in the streaming K1b (no spills) against 16 (same source) and the build before the decision bits
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q > gpurun_out/r4/s18_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s18_tests.txt && rc=99
tail -5 gpurun_out/r4/s18_tests.txt
[ $rc -ne 0 ] && exit $rc
for r in 1 2; do for v in main k1s1024 k1b_before; do
  if [ $v = main ]; then lib=""; else lib="FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so"; fi
  env $lib python3 bench.py --steps 40 --no-e2e --no-cpu --no-devpath $( [ $v = k1b_before ] && echo --draw-words ) 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])"
done; done 2>&1 | tee gpurun_out/r4/s18_ab_12waves.txt
