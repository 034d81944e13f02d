#!/bin/bash
# profiles of record (round 4) + the bench line of the same build
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1100 bash tools/profile_round.sh r4_c3 > gpurun_out/r4/s12_profile_round.txt 2>&1; rc=$?
tail -5 gpurun_out/r4/s12_profile_round.txt
cp profiles/r4_c3_* gpurun_out/r4/ 2>/dev/null; cp profiles/traffic.json gpurun_out/r4/ 2>/dev/null
cat gpurun_out/prof_r4_c3/bench.err 2>/dev/null | tail -5
exit $rc
