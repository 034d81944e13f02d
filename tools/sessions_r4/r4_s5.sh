#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_multi.py -x -q > gpurun_out/r4/s5_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s5_tests.txt && rc=99
tail -15 gpurun_out/r4/s5_tests.txt
exit $rc
