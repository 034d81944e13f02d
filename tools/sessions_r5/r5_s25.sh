#!/bin/bash
# round 5, session 25: the matrix rows land in every pinned decoder slot, not only the first (finish of a 16 M-row matrix)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_multi.py -m gpu -x -q > $O/s25_tests.txt 2>&1
tail -3 $O/s25_tests.txt
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_LEND_ROWS=0" "FASTF_X=1" > $O/s25_windows.txt 2>&1
grep -v phases $O/s25_windows.txt | cut -c1-520
