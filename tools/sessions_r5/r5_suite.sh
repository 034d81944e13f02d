#!/bin/bash
# round 5: the whole GPU suite in one process
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; rc=$?; grep -q "Memory access fault" $O/suite.txt && rc=99
tail -6 $O/suite.txt
exit $rc
