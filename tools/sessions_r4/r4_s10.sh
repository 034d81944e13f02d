#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > gpurun_out/r4/s10_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s10_tests.txt && rc=99
tail -3 gpurun_out/r4/s10_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 tools/ab_lib.sh 2 "main" "main FASTF_K3_PER_CU=4" > gpurun_out/r4/s10_ab.txt 2>&1
cut -c1-330 gpurun_out/r4/s10_ab.txt
