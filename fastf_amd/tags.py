"""Python mirror of the tag-histogram seam (crb / extract, include/fastf_amd.h §2c) for tests and tools."""
import ctypes as C

import numpy as np

from . import _lib
from .engine import _libc_free


class KeyDict:
    """exact string <-> 64-bit key dictionary in its registering (intern) mode"""

    def __init__(self):
        self.h = _lib.lib().fastf_keydict_create()

    def intern(self, s: bytes) -> int:
        return int(_lib.lib().fastf_keydict_intern(self.h, s, len(s)))

    def intern_many(self, strings) -> np.ndarray:
        return np.array([self.intern(s) for s in strings], dtype=np.uint64)

    def decode(self, key: int) -> bytes:
        buf = C.create_string_buffer(4096)
        n = _lib.lib().fastf_keydict_decode(self.h, int(key), buf, 4096)
        if n < 0:
            raise _lib.FastfError("undecodable key %#x" % int(key))
        return buf.raw[:n]

    def close(self):
        if self.h:
            _lib.lib().fastf_keydict_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TagHist:
    """device tag histogram: distinct values, counts and first-occurrence record index"""

    def __init__(self, device=0):
        h = C.c_void_p()
        _lib.check(_lib.lib().fastf_taghist_create(device, C.byref(h)))
        self.h = h

    def push(self, key1, key2=None):
        k1 = np.ascontiguousarray(key1, dtype=np.uint64)
        k2 = None if key2 is None else np.ascontiguousarray(key2, dtype=np.uint64)
        _lib.check(_lib.lib().fastf_taghist_push(self.h, k1.ctypes.data, None if k2 is None else k2.ctypes.data, len(k1)))

    def finish(self):
        r = _lib.TagHistResult()
        _lib.check(_lib.lib().fastf_taghist_finish(self.h, C.byref(r)))

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dtype=dt)
        out = dict(n_records=int(r.n_records), n_valid=int(r.n_valid), n_key1_present=int(r.n_key1_present),
                   key1=arr(r.key1, r.n1, np.uint64), count1=arr(r.count1, r.n1, np.uint64), first1=arr(r.first1, r.n1, np.uint64))
        if r.n_pairs:
            out.update(pair_k1=arr(r.pair_k1, r.n_pairs, np.uint32), pair_key2=arr(r.pair_key2, r.n_pairs, np.uint64),
                       pair_count=arr(r.pair_count, r.n_pairs, np.uint64), pair_first=arr(r.pair_first, r.n_pairs, np.uint64))
        return out

    def close(self):
        if self.h:
            _lib.lib().fastf_taghist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def crb_text(bam_path: str):
    """decompressed bytes of `fastF crb -b bam` (+ number of records)"""
    p, n, nr = C.c_void_p(), C.c_size_t(), C.c_uint64()
    _lib.check(_lib.lib().fastf_crb_text(bam_path.encode(), C.byref(p), C.byref(n), C.byref(nr)))
    txt = C.string_at(p.value, n.value)
    _libc_free(p)
    return txt, int(nr.value)


def extract_text(bam_path: str, tag: bytes, type_: int = 0):
    """bytes of tag_summary.csv of `fastF extract -b bam -t tag -T type` (+ records, valid)"""
    p, n, nr, nv = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
    _lib.check(_lib.lib().fastf_extract_text(bam_path.encode(), tag, type_, C.byref(p), C.byref(n), C.byref(nr), C.byref(nv)))
    txt = C.string_at(p.value, n.value)
    _libc_free(p)
    return txt, int(nr.value), int(nv.value)
