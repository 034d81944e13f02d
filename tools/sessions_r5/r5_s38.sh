#!/bin/bash
# round 5, session 38: two windows in flight again, now that the scout is not the period
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_BAM_EARLY=1" "FASTF_X=1" "FASTF_BAM_EARLY=1" > $O/s38_windows.txt 2>&1
grep -v "phases" $O/s38_windows.txt | cut -c1-420
