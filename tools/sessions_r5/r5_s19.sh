#!/bin/bash
# round 5, session 19: keys wider than 64 bits through the sharded host path (ranks sharing the one GPU over gloo) and the 64-bit refusals
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/s19_tests.txt 2>&1
tail -15 $O/s19_tests.txt
