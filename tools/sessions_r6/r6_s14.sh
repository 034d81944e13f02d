#!/bin/bash
# round 6, session 14: the records of the round's last state — the GPU suite as the driver runs it, smoke, bench.py as the driver
# runs it, the rocprofv3 passes behind profiles/r6_c3_*
set -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/s14_suite.txt 2>&1; rc=$?
echo "suite rc $rc" >> $O/s14_suite.txt
grep -E "passed|failed|suite rc" $O/s14_suite.txt | tail -3
[ $rc -ne 0 ] && { grep -v "^\[gpu unit\].*rc=0" $O/s14_suite.txt | tail -60; exit $rc; }
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/s14_smoke.txt 2>&1 || { tail -5 $O/s14_smoke.txt; exit 1; }
tail -1 $O/s14_smoke.txt
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/s14_bench_default.json 2> $O/s14_bench_default.err; rc=$?
t1=$(date +%s.%N)
echo "bench rc $rc, wall $(python3 -c "print(round($t1-$t0,1))") s"
[ $rc -ne 0 ] && { tail -20 $O/s14_bench_default.err; exit $rc; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/s14_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], round(d["roofline"]["frac"],3), d.get("parity_vs_cpu"), round(d["whole_path"]["read_frac_of_peak"],3))
print({k:(round(v/1e6,1) if v else v) for k,v in d["scopes"]["e2e_records_per_s_to_process_exit"].items()})
c=d["e2e"].get("cell_ranger_shaped_200M_cold") or {}
print({k:c.get(k) for k in ("value","seconds","bam_bytes_per_s","filesystem","cached_fraction_before_the_run","same_matrix_rows_as_the_cached_run")})
print(json.dumps(d["step_with_draw_generation"])[:200])
PY
bash tools/profile_round.sh r6_c3 > $O/s14_profile.log 2>&1; echo "profile rc $?"; tail -2 $O/s14_profile.log
