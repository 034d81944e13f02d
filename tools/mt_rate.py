"""diagnostic: draws per second of mt_fill_kernel, words form and decision-bit form (one launch of N draws, wall clock incl. sync)"""
import ctypes as C, time, sys
import numpy as np
import torch
from fastf_amd import _lib
L = _lib.lib()
L.fastf_debug_mt_fill.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
L.fastf_debug_mt_fill_bits.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
torch.zeros(1, device="cuda")
for n in (1 << 22, 1 << 24, 1 << 26):
    cs = np.asarray([n], np.uint64)
    out = np.zeros(max(n, (1 << 27) // 32), np.uint32)          # words form: n words; bits form: the ring's 2^27 / 32 words
    for form in ("words", "bits"):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            if form == "words":
                assert L.fastf_debug_mt_fill(0, 926, 0, cs.ctypes.data, 1, out.ctypes.data) == 0
            else:
                assert L.fastf_debug_mt_fill_bits(0, 926, 0, 0, cs.ctypes.data, 1, 1 << 31, 1 << 27, out.ctypes.data) == 0
            best = min(best, time.perf_counter() - t)
        print(form, n, "draws: %.2f ms (incl. alloc, copies back)" % (best * 1e3), flush=True)
