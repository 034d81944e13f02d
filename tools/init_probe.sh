#!/bin/bash
# diagnostic: time of the first HIP stream creation (runtime + device context) of a bare process under a few runtime settings
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build/exit_probe tools/exit_probe.hip || exit 1
python3 - <<'PY'
import os, subprocess, time
variants = [{}, {"HSA_ENABLE_INTERRUPT": "0"}, {"GPU_MAX_HW_QUEUES": "1"}, {"HSA_ENABLE_SDMA": "0"}, {"HIP_INITIAL_DM_SIZE": "0"},
            {"HSA_NO_SCRATCH_RECLAIM": "1"}, {"HSA_DISABLE_FRAGMENT_ALLOCATOR": "1"}, {"ROCR_VISIBLE_DEVICES": "0"}, {"HSA_TOOLS_LIB": ""},
            {"HSA_ENABLE_INTERRUPT": "0", "GPU_MAX_HW_QUEUES": "1", "HIP_INITIAL_DM_SIZE": "0"}]
for rep in range(3):
    for v in variants:
        env = dict(os.environ, **v)
        t0 = time.perf_counter()
        out = subprocess.run(["build/exit_probe", "0", "0", "-", "0"], capture_output=True, text=True, env=env).stdout.strip()
        print("%-70s wall %.3f s   %s" % (" ".join("%s=%s" % kv for kv in v.items()) or "(default)", time.perf_counter() - t0, out), flush=True)
PY
