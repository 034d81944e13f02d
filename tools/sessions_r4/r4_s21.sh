#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4/s21_gpu_suite.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s21_gpu_suite.txt && rc=99
tail -8 gpurun_out/r4/s21_gpu_suite.txt
exit $rc
