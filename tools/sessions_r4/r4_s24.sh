#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q > gpurun_out/r4/s24_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s24_tests.txt && rc=99
tail -3 gpurun_out/r4/s24_tests.txt
[ $rc -ne 0 ] && exit $rc
for v in main; do
  if [ $v = main ]; then lib=""; else lib="FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so"; fi
  echo "== $v" | tee -a gpurun_out/r4/s24_mt_rate.txt
  env $lib timeout -k 10 200 python3 tools/mt_rate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/s24_mt_rate.txt
  for hd in 0 1; do env $lib FASTF_HOST_DRAWS=$hd timeout -k 10 300 python3 bench.py --steps 5 --no-e2e --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host_draws=$hd', d['device_path']['value'], d['device_path']['h2d_GBs'], d['device_path']['runs_s'])" | tee -a gpurun_out/r4/s24_mt_rate.txt; done
done
