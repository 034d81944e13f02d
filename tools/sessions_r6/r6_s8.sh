#!/bin/bash
# round 6, session 8: the GPU suite in its new form (child process per file / heavy test), the way the driver runs it
set -o pipefail
mkdir -p gpurun_out/r6
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r6/suite_s8.txt 2>&1
rc=$?
echo "suite rc $rc" >> gpurun_out/r6/suite_s8.txt
grep -E "^\[gpu unit\]|passed|failed|suite rc" gpurun_out/r6/suite_s8.txt | tail -40
exit 0
