#!/bin/bash
# round 5, session 24: hop chains that do not meet are repaired on the device (no window goes back to the host for a false guess)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_records.py tests/test_gpu_e2e.py tests/test_gpu_inflate.py -m gpu -x -q > $O/s24_tests.txt 2>&1
tail -12 $O/s24_tests.txt
