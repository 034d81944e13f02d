#!/bin/bash
# round 6, session 16: the GPU suite of the round's LAST state, in ONE process (FASTF_TEST_INPROCESS=1) under tools/pinwatch.c, minus the
# two places where RCCL comes up: how many pageable copies of a megabyte or more are left, and what the ledger says
set -o pipefail
gcc -O2 -g -shared -fPIC -o build/libpinwatch.so tools/pinwatch.c -ldl
rm -rf gpurun_out/r6/pinwatch_final; mkdir -p gpurun_out/r6/pinwatch_final
export PINWATCH_LOG=$PWD/gpurun_out/r6/pinwatch_final/log
FASTF_TEST_INPROCESS=1 LD_PRELOAD=$PWD/build/libpinwatch.so timeout -k 10 1000 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider \
   --deselect tests/test_gpu_dist.py::test_collectives_on_rccl_with_one_rank --ignore tests/test_gpu_multi.py > gpurun_out/r6/suite_pinwatch_final.txt 2>&1
echo "suite rc $?" >> gpurun_out/r6/suite_pinwatch_final.txt
tail -3 gpurun_out/r6/suite_pinwatch_final.txt
grep -h "pinwatch summary" gpurun_out/r6/pinwatch_final/* | cut -c1-420 | sort | uniq -c | sort -rn | head -12
grep -h -E "^(FREE-WHILE|STALE|REGISTER-OVERLAP|UNREGISTER-UNKNOWN)" gpurun_out/r6/pinwatch_final/* | awk '{print $1}' | sort | uniq -c
grep -h "big pageable copies from" gpurun_out/r6/pinwatch_final/* | sort | uniq -c | sort -rn | head
exit 0
