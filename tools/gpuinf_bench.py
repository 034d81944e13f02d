"""diagnostic: device BGZF inflate of the first blocks of a BAM (kernel + copies), GB/s of inflated bytes
usage: tools/gpuinf_bench.py in.bam [max_blocks]"""
import ctypes as C, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from fastf_amd import _lib


class Blk(C.Structure):
    _fields_ = [("coff", C.c_uint64), ("clen", C.c_uint32), ("isize", C.c_uint32), ("uoff", C.c_uint64)]


path = sys.argv[1]; maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
data = np.fromfile(path, dtype=np.uint8, count=1200 << 20)
raw = data.tobytes()
L0 = _lib.lib()
L0.fastf_pinned_alloc.restype = C.c_void_p
pin = L0.fastf_pinned_alloc(len(raw) + 4096)                       # the driver takes pinned compressed bytes
C.memmove(pin, raw, len(raw)); C.memset(pin + len(raw), 0, 4096)
blks = []; pos = 0; uoff = 0
while pos + 18 < len(raw) and len(blks) < maxb:
    xlen = struct.unpack_from("<H", raw, pos + 10)[0]
    bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
    if pos + bsize > len(raw): break
    isize = struct.unpack_from("<I", raw, pos + bsize - 4)[0]
    blks.append((pos + 12 + xlen, bsize - 12 - xlen - 8, isize, uoff)); uoff += isize; pos += bsize
n = len(blks)
desc = (Blk * n)(*[Blk(*b) for b in blks])
L = _lib.lib()
L.fastf_gpuinf_create.restype = C.c_void_p; L.fastf_gpuinf_create.argtypes = [C.c_int]
L.fastf_gpuinf_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
out_p = L.fastf_pinned_alloc(uoff + 64); status = (C.c_uint8 * n)()
g = L.fastf_gpuinf_create(0)
cbuf = pin
for rep in range(4):
    t = time.perf_counter(); rc = L.fastf_gpuinf_run(g, cbuf, desc, n, out_p, status); dt = time.perf_counter() - t
    print("run %d: rc %d, %d blocks, %.1f MB in, %.1f MB out, %.1f ms -> %.1f GB/s inflated; declined %d" % (rep, rc, n, pos / 1e6, uoff / 1e6, dt * 1e3, uoff / dt / 1e9, sum(1 for s in status if s)))

if hasattr(L, "fastf_debug_gi2_stamps"):                      # experiment build with FASTF_X_GI2_STAMPS: cycles per section of the symbol loop
    try:
        st = (C.c_ulonglong * 8)()
        if L.fastf_debug_gi2_stamps(st) == 0:
            tot = float(sum(st)) or 1.0
            names = ["headers+construction", "loop head+epochs", "refill", "lit/len symbol", "literal staged", "len extra+dist symbol+extra", "token staged", "final flush"]
            print("symbol-loop stamps (share of one lane-0 wave clock, all waves summed): " + "; ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, st)))
    except Exception as ex:
        print("stamps:", ex)
