#!/bin/bash
# K3h with head / copy / distinct flags as scalar lane masks against the build before (build/k1b_before = HEAD)
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > gpurun_out/r4/s32_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s32_tests.txt && rc=99
tail -3 gpurun_out/r4/s32_tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/ab_k.sh 4 main k1b_before
