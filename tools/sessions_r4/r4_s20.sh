#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/kernel_pmc.sh filter_pack_stream > gpurun_out/r4/s20_pmc_k1b.txt 2>&1
cat gpurun_out/r4/s20_pmc_k1b.txt
