#!/bin/bash
# round 5, session 21: where FASTF_DEVICES=0,0 spends more than one device (stage lines of the CLI), 80 M records
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1000 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_DEVICES=0,0" > $O/s21_windows.txt 2>&1
cat $O/s21_windows.txt | cut -c1-600
