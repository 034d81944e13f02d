#!/bin/bash
# round 5, session 17: UMIs of up to 32 bases (sub-groups in the sorted word), two windows in flight behind its knob
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_records.py tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -x -q > $O/s17_tests.txt 2>&1
tail -15 $O/s17_tests.txt
