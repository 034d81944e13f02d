#!/bin/bash
# round 5, session 37: the scout maps a window's pages with three helpers, the device's share starts at 0.85: reader tests, window trace, windows
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_records.py tests/test_gpu_inflate.py tests/test_gpu_e2e.py -m gpu -x -q > $O/s37_tests.txt 2>&1; rc=$?
tail -2 $O/s37_tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/e2e_one.sh 80000000 91 > $O/s37_e2e_one.txt 2>&1
grep "scout:\|window [0-9]*:\|lists" $O/s37_e2e_one.txt | cut -c1-250
timeout -k 10 600 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_X=1" > $O/s37_windows.txt 2>&1
grep -v "phases" $O/s37_windows.txt | cut -c1-420
