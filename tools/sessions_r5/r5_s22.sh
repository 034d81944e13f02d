#!/bin/bash
# round 5, session 22: multi-device finish with huge-page row buffers (no vectors): tests, then one device against FASTF_DEVICES=0,0
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/s22_tests.txt 2>&1
tail -3 $O/s22_tests.txt
timeout -k 10 1000 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_DEVICES=0,0" "FASTF_X=1" "FASTF_DEVICES=0,0" > $O/s22_windows.txt 2>&1
cat $O/s22_windows.txt | cut -c1-600
