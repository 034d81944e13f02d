#!/bin/bash
# round 5, session 15: two windows in flight (second device context, the next window's device share queued early)
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_records.py tests/test_gpu_e2e.py tests/test_gpu_tags.py tests/test_gpu_parity.py -x -q > $O/s15_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s15_tests.txt && rc=99
tail -3 $O/s15_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_BAM_EARLY=0" "FASTF_X=0" > $O/s15_windows.txt 2>&1
cat $O/s15_windows.txt
