#!/bin/bash
# round-4 session 2: blocked record layout (parity, A/B against the SoA form), pitched H2D copies
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_dist.py -x -q > gpurun_out/r4/s2_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s2_tests.txt && rc=99
tail -5 gpurun_out/r4/s2_tests.txt
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
timeout -k 10 120 tools/bin/h2d_2d_probe > gpurun_out/r4/s2_h2d_2d.txt 2>&1; cat gpurun_out/r4/s2_h2d_2d.txt
AB_ARGS="" timeout -k 10 300 tools/ab_lib.sh 2 "main" > gpurun_out/r4/s2_ab_blocked.txt 2>&1 || exit 1
AB_ARGS="--soa" timeout -k 10 300 tools/ab_lib.sh 2 "main" >> gpurun_out/r4/s2_ab_blocked.txt 2>&1 || exit 1
cat gpurun_out/r4/s2_ab_blocked.txt
