#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 tools/ab_lib.sh 2 "main" "main FASTF_K3_PER_CU=3" "k3h6 FASTF_K3_PER_CU=3" "k3h6c FASTF_K3_PER_CU=3" "k3h6c FASTF_K3_PER_CU=4" > gpurun_out/r4/s9_ab_k3h.txt 2>&1
cut -c1-330 gpurun_out/r4/s9_ab_k3h.txt
timeout -k 10 500 tools/kernel_pmc.sh filter_pack_stream > gpurun_out/r4/s9_pmc_k1b_blocked.txt 2>&1
cat gpurun_out/r4/s9_pmc_k1b_blocked.txt
