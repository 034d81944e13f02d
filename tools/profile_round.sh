#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs
# (counters are never combined with tracing).  Run on the GPU box from the repo root:  tools/profile_round.sh r2_c3
# The profiled command is bench.py on its default workload (BASELINE configs[2], 200 M records), device steps only.
set -e
tag=${1:-r4_c3}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
B="bench.py --no-cpu --no-e2e --no-devpath"
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $B --steps 5 --warmup 1 > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 $B --steps 2 --warmup 1 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 $B --steps 2 --warmup 1 > $out/write.log 2>&1
keys=$(grep '^{"metric"' $out/stats.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['counters']['keys'])")
recs=$(grep '^{"metric"' $out/stats.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['counters']['total'])")
python3 tools/prof_summary.py $tag $out/stats $out/fetch $out/write $keys $recs > $out/summary.json
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_rocprofv3_kernel_stats_raw.csv
# the bench line of the PROFILED command itself: its HIP-event kernel times are the ones the rocprofv3 averages must agree with
# (kernels run a few per cent slower under the tracer than in the plain run below)
grep '^{"metric"' $out/stats.log | tail -1 > profiles/${tag}_bench_line_under_rocprofv3.json
python3 bench.py --steps 100 --warmup 3 > profiles/${tag}_bench_line.json 2> $out/bench.err
# (only gpurun_out/ travels back from the GPU box: a copy of everything this script put under profiles/)
mkdir -p $out/profiles && cp profiles/${tag}_* $out/profiles/
echo "profiles/${tag}_* written (copies under $out/profiles/)"
