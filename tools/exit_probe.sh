#!/bin/bash
# process-exit cost by what the process holds (GPU box): tools/exit_probe.sh [file-to-map]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build/exit_probe tools/exit_probe.hip || exit 1
f=${1:--}
python3 - "$f" <<'PY'
import subprocess, sys, time
f = sys.argv[1]
cases = [("0", "0", "-", "0"), ("0", "0", "-", "2"), ("4096", "256", "-", "0"), ("4096", "256", "-", "2"), ("4096", "0", "-", "0"), ("4096", "0", "-", "1"), ("0", "1024", "-", "0"), ("0", "1024", "-", "1"),
         ("0", "0", f, "0"), ("0", "0", f, "1"), ("4096", "1024", f, "0"), ("4096", "1024", f, "1")]
for rep in range(2):
    for c in cases:
        t0 = time.perf_counter()
        out = subprocess.run(["build/exit_probe", *c], capture_output=True, text=True).stdout.strip()
        print("dev %5s MiB  pinned %5s MiB  map %-18s free_first %s   wall %.3f s   %s" % (c[0], c[1], c[2][-18:], c[3], time.perf_counter() - t0, out), flush=True)
PY
