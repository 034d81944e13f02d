#!/bin/bash
# round 5, session 18: keys wider than 64 bits through the multi-device engine (devices aliased on the one GPU)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -x -q > $O/s18_tests.txt 2>&1
tail -15 $O/s18_tests.txt
