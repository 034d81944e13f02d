#!/bin/bash
# round 5, session 12: the decode kernel reading the compressed bytes out of pinned host memory (no H2D copy), window sizes
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 bash tools/e2e_windows.sh 80000000 91 "FASTF_X=0" "FASTF_GI_ZEROCOPY=1" "FASTF_BAM_WINDOW=805306368" "FASTF_BAM_WINDOW=805306368 FASTF_GI_ZEROCOPY=1" > $O/s12_windows.txt 2>&1
cat $O/s12_windows.txt
