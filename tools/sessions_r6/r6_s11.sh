#!/bin/bash
# round 6, session 11: (1) does GPU_PINNED_MIN_XFER_SIZE=1000000 switch the on-the-fly pinning off (pageable copy rates, both runtimes),
# (2) bench.py as the driver runs it (with the cold-cache e2e line), (3) the rocprofv3 passes behind profiles/r6_c3_*
set -o pipefail
O=gpurun_out/r6; mkdir -p $O
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
{ echo "== GPU_PINNED_MIN_XFER_SIZE=1000000, /opt/rocm runtime"; GPU_PINNED_MIN_XFER_SIZE=1000000 timeout -k 10 120 tools/bin/pin_probe facts 64 | sed -n '/^4\./,$p';
  echo "== GPU_PINNED_MIN_XFER_SIZE=1000000, torch's runtime"; GPU_PINNED_MIN_XFER_SIZE=1000000 LD_PRELOAD=$TL/libamdhip64.so timeout -k 10 120 tools/bin/pin_probe facts 64 | sed -n '/^4\./,$p'; } > $O/pin_probe_env.txt 2>&1
cat $O/pin_probe_env.txt
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/s11_bench_default.json 2> $O/s11_bench_default.err; rc=$?
t1=$(date +%s.%N)
echo "bench rc $rc, wall $(python3 -c "print(round($t1-$t0,1))") s"
[ $rc -ne 0 ] && { tail -20 $O/s11_bench_default.err; exit $rc; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/s11_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], round(d["roofline"]["frac"],3), d.get("parity_vs_cpu"), round(d["whole_path"]["read_frac_of_peak"],3))
print({k:(round(v/1e6,1) if v else v) for k,v in d["scopes"]["e2e_records_per_s_to_process_exit"].items()})
print(json.dumps(d["e2e"].get("cell_ranger_shaped_200M_cold"))[:1200])
print(json.dumps(d["step_with_draw_generation"])[:300])
PY
bash tools/profile_round.sh r6_c3 > $O/s11_profile.log 2>&1; echo "profile rc $?"; tail -3 $O/s11_profile.log
