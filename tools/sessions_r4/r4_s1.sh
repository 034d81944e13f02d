#!/bin/bash
# round-4 session 1: new K3 kernel (tests, A/B against round 3's hash mode), blocked-layout stream probe
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > gpurun_out/r4/s1_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r4/s1_tests.txt
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
timeout -k 10 120 tools/bin/hbm_probe_streams > gpurun_out/r4/s1_probe_streams.txt 2>&1 || exit 1
tail -30 gpurun_out/r4/s1_probe_streams.txt
timeout -k 10 400 tools/ab_lib.sh 2 "main" "main FASTF_K3_OLD_HASH=1" > gpurun_out/r4/s1_ab_k3.txt 2>&1 || exit 1
cat gpurun_out/r4/s1_ab_k3.txt
timeout -k 10 500 tools/kernel_pmc.sh reduce_hashed > gpurun_out/r4/s1_pmc_reduce_hashed.txt 2>&1
cat gpurun_out/r4/s1_pmc_reduce_hashed.txt
