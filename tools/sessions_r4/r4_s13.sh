#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r4/s13_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s13_tests.txt && rc=99
tail -3 gpurun_out/r4/s13_tests.txt
[ $rc -ne 0 ] && exit $rc
for v in "" "FASTF_HOST_DRAWS=1"; do
env $v python3 bench.py --steps 10 --no-e2e --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['device_path']; print('[$v]', {k:p[k] for k in ('value','seconds','push_s','finish_s','h2d_GBs','runs_s','same_result_as_resident_steps')})"
done
python3 - <<'PY'
import ctypes as C, time, numpy as np, sys
sys.path.insert(0,'.')
import torch
from fastf_amd import _lib
L=_lib.lib(); L.fastf_debug_mt_fill.argtypes=[C.c_int,C.c_uint32,C.c_uint64,C.c_void_p,C.c_uint32,C.c_void_p]
n=50_000_000; out=np.zeros(n,np.uint32); cs=np.asarray([n],np.uint64)
for r in range(3):
    t=time.perf_counter(); L.fastf_debug_mt_fill(0,926,0,cs.ctypes.data,1,out.ctypes.data); dt=time.perf_counter()-t
print("mt_fill_kernel: %d draws incl. D2H of %d MB in %.1f ms" % (n, n*4>>20, dt*1e3))
PY
