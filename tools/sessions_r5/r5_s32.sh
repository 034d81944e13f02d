#!/bin/bash
# round 5, session 32: gr_pack_kernel parses its segment out of LDS (staged with 16-byte loads): tests, then the windows of the 80 M-record file
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_records.py tests/test_gpu_e2e.py -m gpu -x -q > $O/s32_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/s32_tests.txt && rc=99
tail -3 $O/s32_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_X=1" "FASTF_GPU_INFLATE_PIN=1 FASTF_GPU_INFLATE_SHARE=0.95 FASTF_GPU_INFLATE_MAX=0.95" > $O/s32_windows.txt 2>&1
grep -v "phases" $O/s32_windows.txt | cut -c1-420
