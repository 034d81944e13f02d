#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs
# (counters are never combined with tracing).  Run on the GPU box from the repo root:  tools/profile_round.sh r1b
set -e
tag=${1:-r1}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $out/write.log 2>&1
python3 tools/prof_summary.py $tag $out/stats $out/fetch $out/write 10000000 > $out/summary.json
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_rocprofv3_kernel_stats_raw.csv
python3 bench.py --steps 20 --warmup 3 > profiles/${tag}_bench_line.json 2> $out/bench.err
echo "profiles/${tag}_* written"
