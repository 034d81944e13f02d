#!/bin/bash
# round 5, session 20: the multi-device engine with the parallel draw generator; then the CLI on one device against
# FASTF_DEVICES=0,0 (two shards aliased on the one GPU) and against wide keys (FASTF_UMI_MAX_BASES=24: the tile-form K1b with
# values, the sort with values) on the 80 M-record Cell-Ranger-shaped file
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/s20_tests.txt 2>&1
tail -5 $O/s20_tests.txt
timeout -k 10 1000 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_DEVICES=0,0" "FASTF_UMI_MAX_BASES=24" > $O/s20_windows.txt 2>&1
cat $O/s20_windows.txt
