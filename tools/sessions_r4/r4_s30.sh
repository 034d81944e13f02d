#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_records.py -x -q > gpurun_out/r4/s30_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s30_tests.txt && rc=99
tail -3 gpurun_out/r4/s30_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 bash tools/gpuinf_ab.sh 6000000 16000 main inf_before main inf_before 2>&1 | tee gpurun_out/r4/s30_inflate_ab.txt
