#!/bin/bash
# round 5, session 31: the device's share of a window pinned high (a slice of 21 K blocks and one of 65 K take the decode kernel the same
# 14 ms: a block's serial latency) — does the window period follow, and what does the consumer (host hop + pack of the host's share) say?
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1100 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_GPU_INFLATE_PIN=1 FASTF_GPU_INFLATE_SHARE=0.95 FASTF_GPU_INFLATE_MAX=0.95" "FASTF_GPU_INFLATE_PIN=1 FASTF_GPU_INFLATE_SHARE=1.0 FASTF_GPU_INFLATE_MAX=1.0" "FASTF_X=1" > $O/s31_windows.txt 2>&1
grep -v "phases" $O/s31_windows.txt | cut -c1-420
