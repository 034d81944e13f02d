#!/bin/bash
# decision-bit draw stream: kernel tests, parity, multi-device, then A/B against the build before it
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q > gpurun_out/r4/s17_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s17_tests.txt && rc=99
tail -15 gpurun_out/r4/s17_tests.txt
[ $rc -ne 0 ] && exit $rc
for r in 1 2; do for v in main k1b_before; do
  if [ $v = main ]; then lib=""; else lib="FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so"; fi
  env $lib python3 bench.py --steps 40 --no-e2e --no-cpu $( [ $v = main ] || echo --draw-words ) 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], d.get('device_path_records_per_s'), [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])"
done; done 2>&1 | tee gpurun_out/r4/s17_ab_drawbits.txt
