#!/bin/bash
# round 5, session 23: the whole GPU suite, then the profiles of record again with the final build
set -o pipefail
O=gpurun_out/r5; mkdir -p $O
bash tools/sessions_r5/r5_suite.sh && timeout -k 10 1100 bash tools/profile_round.sh r5_c3 > $O/profile_round.log 2>&1
rc=$?
tail -3 $O/profile_round.log
mkdir -p $O/profiles && cp profiles/r5_c3_* $O/profiles/ 2>/dev/null
cp gpurun_out/prof_r5_c3/summary.json $O/profiles/r5_c3_summary.json 2>/dev/null
exit $rc
