#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q -k "not full_size and not adversarial" > gpurun_out/r4/s7_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s7_tests.txt && rc=99
tail -40 gpurun_out/r4/s7_tests.txt
exit $rc
