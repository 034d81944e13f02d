#!/bin/bash
# round 5, session 35: every block on the device (share pinned at 1.0), with and without two windows in flight
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
P="FASTF_GPU_INFLATE_PIN=1 FASTF_GPU_INFLATE_SHARE=1.0 FASTF_GPU_INFLATE_MAX=1.0"
timeout -k 10 1100 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "$P" "$P FASTF_BAM_EARLY=1" "FASTF_BAM_EARLY=1" > $O/s35_windows.txt 2>&1
grep -v "phases" $O/s35_windows.txt | cut -c1-420
