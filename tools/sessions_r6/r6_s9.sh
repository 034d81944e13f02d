#!/bin/bash
# round 6, session 9: bench.py exactly as the driver runs it, after the host-memory changes of this round
O=gpurun_out/r6; mkdir -p $O
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/s9_bench_default.json 2> $O/s9_bench_default.err; rc=$?
t1=$(date +%s.%N)
echo "rc $rc, wall $(python3 -c "print(round($t1-$t0,1))") s"
tail -3 $O/s9_bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/s9_bench_default.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["cpu_baseline"]["value"]), d["vs_baseline"], d["config"])
print(d.get("parity_vs_cpu"))
print(round(d["whole_path"]["read_frac_of_peak"],3))
print({k:round(v/1e6,1) for k,v in d["scopes"]["e2e_records_per_s_to_process_exit"].items()})
PY
exit $rc
