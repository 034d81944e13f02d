"""Host-side mirror of the engine C ABI (include/fastf_amd.h).

Nothing here computes results: every call lands in libfastf_amd.so (HIP kernels on the
device, C on the host).  If the library or a HIP device is missing the calls raise.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Batch, Coo, EngineConfig, FastfError, ListsStruct, MT, UmiRows, check

HAS_CB, HAS_XF, HAS_GX, HAS_UB = 1, 2, 4, 8


def draw_threshold(rate_depth: float) -> int:
    """Integer form of bam2db_ds.c:385-390: a record is kept iff draw < threshold."""
    return int(_lib.lib().fastf_draw_threshold(C.c_float(rate_depth)))


def mt_draws(seed: int, skip: int, n: int) -> np.ndarray:
    """n MT19937 draws after init_genrand(seed) and `skip` discarded draws."""
    L = _lib.lib()
    mt = MT()
    L.fastf_mt_seed(C.byref(mt), seed)
    L.fastf_mt_skip(C.byref(mt), skip)
    out = np.empty(n, dtype=np.uint32)
    L.fastf_mt_fill(C.byref(mt), out.ctypes.data, n)
    return out


def sample_cells(n_cells: int, rate_cell: float, seed: int):
    L = _lib.lib()
    out = np.zeros(max(n_cells, 1), dtype=np.uint64)
    ns, used = C.c_size_t(0), C.c_uint64(0)
    rc = L.fastf_sample_cells(n_cells, C.c_float(rate_cell), seed, out.ctypes.data, C.byref(ns), C.byref(used))
    if rc:
        raise FastfError("sample size must lie in [0, n_cells]")
    return out[:ns.value].copy(), int(used.value)


class Lists:
    """Barcode + feature lists as the reference builds them (bam2db_ds.c:229-337)."""

    def __init__(self, barcodes: bytes, features: bytes, rate_cell=1.0, seed=926):
        self._L = _lib.lib()
        self._s = ListsStruct()
        check(self._L.fastf_lists_load_mem(barcodes, len(barcodes), features, len(features),
                                           C.c_float(rate_cell), seed, C.byref(self._s)))
        s = self._s
        self.n_lines_barcodes = s.n_lines_barcodes
        self.n_sampled_target = s.n_sampled_target
        self.n_cells = s.n_cells
        self.n_features = s.n_features
        self.mt_skip = int(s.mt_skip)
        self.dup_barcodes, self.dup_features = s.dup_barcodes, s.dup_features
        self.cell_keys = np.array([s.cell_key[i] for i in range(s.n_cells)], dtype=np.uint64)
        self.feature_keys = np.array([s.feature_key[i] for i in range(s.n_features)], dtype=np.uint64)
        self.barcodes = [s.barcode[i] for i in range(s.n_cells)]
        self.features = [(s.feat_id[i], s.feat_name[i], s.feat_type[i]) for i in range(s.n_features)]

    @property
    def cell_dict(self):
        return self._s.cell_dict

    @property
    def feat_dict(self):
        return self._s.feat_dict

    def barcodes_text(self) -> bytes:
        return b"".join(b + b"\n" for b in self.barcodes)

    def features_text(self) -> bytes:
        return b"".join(b"\t".join(f) + b"\n" for f in self.features)

    def close(self):
        if self._s is not None:
            self._L.fastf_lists_free(C.byref(self._s))
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_records(lists: Lists, flags, xf, cb, gx, ub, long_umis=False):
    """String-level records → packed SoA via the product's own packer (host_io.c).  long_umis: UMIs of up to 32 bases, a
    fifth array (bases 17..) is returned behind the four"""
    L = _lib.lib()
    if long_umis:
        L.fastf_pack_records_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fastf_pack_records_ext.restype = None
        flags = np.ascontiguousarray(flags, dtype=np.uint8); xf = np.ascontiguousarray(xf, dtype=np.int32)
        cb, gx, ub = (np.ascontiguousarray(a) for a in (cb, gx, ub))
        n = len(flags)
        out = (np.empty(n, np.uint64), np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32))
        L.fastf_pack_records_ext(lists.cell_dict, lists.feat_dict, n, flags.ctypes.data, xf.ctypes.data, cb.ctypes.data, cb.dtype.itemsize,
                                 gx.ctypes.data, gx.dtype.itemsize, ub.ctypes.data, ub.dtype.itemsize, *(a.ctypes.data for a in out))
        return out
    L.fastf_pack_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fastf_pack_records.restype = None
    flags = np.ascontiguousarray(flags, dtype=np.uint8)
    xf = np.ascontiguousarray(xf, dtype=np.int32)
    cb, gx, ub = (np.ascontiguousarray(a) for a in (cb, gx, ub))
    n = len(flags)
    cb_key = np.empty(n, dtype=np.uint64)
    gx_key = np.empty(n, dtype=np.uint64)
    umi = np.empty(n, dtype=np.uint32)
    meta = np.empty(n, dtype=np.uint32)
    L.fastf_pack_records(lists.cell_dict, lists.feat_dict, n, flags.ctypes.data, xf.ctypes.data,
                         cb.ctypes.data, cb.dtype.itemsize, gx.ctypes.data, gx.dtype.itemsize,
                         ub.ctypes.data, ub.dtype.itemsize,
                         cb_key.ctypes.data, gx_key.ctypes.data, umi.ctypes.data, meta.ctypes.data)
    return cb_key, gx_key, umi, meta


class PinnedBatch:
    """n packed records in pinned host memory (fastf_pinned_alloc): cb_key u64 | gx_key u64 | umi u32 | meta u32,
    as numpy views.  What fastf_engine_push_pinned sends to the device without a staging copy."""

    def __init__(self, n):
        self._L = _lib.lib()
        self.n = int(n)
        self._p = self._L.fastf_pinned_alloc(max(self.n, 1) * 24)
        if not self._p:
            raise FastfError(self._L.fastf_last_error().decode())
        buf = (C.c_uint8 * (max(self.n, 1) * 24)).from_address(self._p)
        raw = np.frombuffer(buf, dtype=np.uint8)
        self.cb_key = raw[:8 * n].view(np.uint64)
        self.gx_key = raw[8 * n:16 * n].view(np.uint64)
        self.umi = raw[16 * n:20 * n].view(np.uint32)
        self.meta = raw[20 * n:24 * n].view(np.uint32)

    def fill(self, off, cb, gx, umi, meta):
        """copy torch device tensors (int64/int32 bit patterns) or numpy arrays into [off, off + len)"""
        def put(dst, src, dt):
            if hasattr(src, "cpu"):
                import torch
                view = torch.from_numpy(dst[off:off + src.numel()].view(dt))
                view.copy_(src)                       # device -> pinned host, no intermediate
            else:
                dst[off:off + len(src)] = src
        put(self.cb_key, cb, np.int64); put(self.gx_key, gx, np.int64)
        put(self.umi, umi, np.int32); put(self.meta, meta, np.int32)

    def batch(self, a=0, b=None):
        b = self.n if b is None else b
        return Batch(self.cb_key[a:b].ctypes.data, self.gx_key[a:b].ctypes.data, self.umi[a:b].ctypes.data,
                     self.meta[a:b].ctypes.data, b - a)

    def close(self):
        if getattr(self, "_p", None):
            self.cb_key = self.gx_key = self.umi = self.meta = None
            self._L.fastf_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """fastf_engine_* (host buffers) and fastf_dev_* (device pointers) of include/fastf_amd.h."""

    def __init__(self, cell_keys, feature_keys, rate_depth=1.0, seed=926, mt_skip=0,
                 umi_max_bases=12, n_shards=1, shard_rank=0, device=0,
                 batch_records=0, key_capacity=0, threshold=None, devices=None):
        self._L = _lib.lib()
        self._cell_keys = np.ascontiguousarray(cell_keys, dtype=np.uint64)
        self._feature_keys = np.ascontiguousarray(feature_keys, dtype=np.uint64)
        cfg = EngineConfig()
        cfg.cell_keys = self._cell_keys.ctypes.data
        cfg.n_cells = len(self._cell_keys)
        cfg.feature_keys = self._feature_keys.ctypes.data
        cfg.n_features = len(self._feature_keys)
        cfg.draw_threshold = draw_threshold(rate_depth) if threshold is None else threshold
        cfg.umi_max_bases = umi_max_bases
        cfg.mt_seed, cfg.mt_skip = seed, mt_skip
        cfg.n_shards, cfg.shard_rank = n_shards, shard_rank
        cfg.device = device
        cfg.batch_records, cfg.key_capacity = batch_records, key_capacity
        if devices is not None:                # one engine driving several devices (ordinals may repeat: one-GPU rehearsal)
            self._devices = np.ascontiguousarray(devices, dtype=np.int32)
            cfg.n_devices, cfg.devices = len(self._devices), self._devices.ctypes.data
        h = C.c_void_p()
        check(self._L.fastf_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.n_shards = n_shards
        cb, fb, ub, tb = (C.c_uint32() for _ in range(4))
        check(self._L.fastf_engine_key_bits(h, C.byref(cb), C.byref(fb), C.byref(ub), C.byref(tb)))
        self.cell_bits, self.feature_bits, self.umi_bits, self.key_bits = cb.value, fb.value, ub.value, tb.value

    @classmethod
    def from_lists(cls, lists: Lists, rate_depth=1.0, seed=926, **kw):
        return cls(lists.cell_keys, lists.feature_keys, rate_depth=rate_depth, seed=seed,
                   mt_skip=lists.mt_skip, **kw)

    # ---- host-buffer streaming API ----
    @staticmethod
    def _batch(cb_key, gx_key, umi, meta, umi_ext=None):
        arrs = (np.ascontiguousarray(cb_key, dtype=np.uint64), np.ascontiguousarray(gx_key, dtype=np.uint64),
                np.ascontiguousarray(umi, dtype=np.uint32), np.ascontiguousarray(meta, dtype=np.uint32),
                None if umi_ext is None else np.ascontiguousarray(umi_ext, dtype=np.uint32))
        b = Batch(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, len(arrs[0]),
                  None if arrs[4] is None else arrs[4].ctypes.data)
        return b, arrs

    def push(self, cb_key, gx_key, umi, meta, draws=None, umi_ext=None):
        """umi_ext: bases 17.. of every UMI (pack_records(..., long_umis=True)) for an engine with umi_max_bases > 16"""
        b, keep = self._batch(cb_key, gx_key, umi, meta, umi_ext)
        if draws is None:
            check(self._L.fastf_engine_push(self._h, C.byref(b)))
        else:
            d = np.ascontiguousarray(draws, dtype=np.uint32)
            check(self._L.fastf_engine_push_draws(self._h, C.byref(b), d.ctypes.data, len(d)))
        del keep

    def push_pinned(self, pb: PinnedBatch, a=0, b=None):
        """zero-copy push of pinned records [a, b); the memory must stay untouched until wait_input() / finish()"""
        bt = pb.batch(a, b)
        check(self._L.fastf_engine_push_pinned(self._h, C.byref(bt)))

    def wait_input(self):
        check(self._L.fastf_engine_wait_input(self._h))

    def lend_rows(self, ptr, nbytes):
        """pinned host memory (address, bytes) the next finish() may place the rows in; 0 / None withdraws the loan"""
        check(self._L.fastf_engine_lend_rows(self._h, ptr or None, nbytes))

    def finish(self, copy=True):
        """copy=False: the arrays are VIEWS of the engine's row buffer (valid until the next reset / finish / close): what a C
        caller of fastf_engine_finish gets, without the 12 bytes per row this wrapper would copy"""
        coo = Coo()
        cnt = (C.c_uint64 * 3)()
        check(self._L.fastf_engine_finish(self._h, C.byref(coo), C.byref(cnt)))
        n = coo.nnz

        def arr(p):
            if not n:
                return np.zeros(0, dtype=np.uint32)
            a = np.ctypeslib.as_array(p, shape=(n,))
            return a.copy() if copy else a
        return dict(feature=arr(coo.feature), cell=arr(coo.cell), count=arr(coo.count), nnz=n,
                    total=int(cnt[0]), sampled=int(cnt[1]), valid=int(cnt[2]))

    def umi_rows(self):
        r = UmiRows()
        check(self._L.fastf_engine_umi_rows(self._h, C.byref(r)))
        n = r.n

        def arr(p, dt=np.uint32):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dtype=dt)
        return dict(feature=arr(r.feature), cell=arr(r.cell), n_copy=arr(r.n_copy), umi=arr(r.umi),
                    nonnull=arr(r.nonnull, np.uint8), n=n)

    def format_matrix(self, res, bam_label: bytes, rate_cell, rate_depth, n_feature, n_barcode) -> bytes:
        f = np.ascontiguousarray(res["feature"], dtype=np.uint32)
        c = np.ascontiguousarray(res["cell"], dtype=np.uint32)
        k = np.ascontiguousarray(res["count"], dtype=np.uint32)
        coo = Coo(f.ctypes.data_as(C.POINTER(C.c_uint32)), c.ctypes.data_as(C.POINTER(C.c_uint32)),
                  k.ctypes.data_as(C.POINTER(C.c_uint32)), len(f))
        cnt = (C.c_uint64 * 3)(res["total"], res["sampled"], res["valid"])
        out, n = C.c_void_p(), C.c_size_t()
        check(self._L.fastf_format_matrix(bam_label, C.c_float(rate_cell), C.c_float(rate_depth), C.byref(cnt),
                                          n_feature, n_barcode, C.byref(coo), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            _libc_free(out)

    def format_umi_rows(self, rows) -> bytes:
        a = {k: np.ascontiguousarray(rows[k]) for k in ("feature", "cell", "n_copy", "umi", "nonnull")}
        r = UmiRows(a["feature"].ctypes.data_as(C.POINTER(C.c_uint32)), a["cell"].ctypes.data_as(C.POINTER(C.c_uint32)),
                    a["n_copy"].ctypes.data_as(C.POINTER(C.c_uint32)), a["umi"].ctypes.data_as(C.POINTER(C.c_uint32)),
                    a["nonnull"].ctypes.data_as(C.POINTER(C.c_uint8)), rows["n"])
        out, n = C.c_void_p(), C.c_size_t()
        check(self._L.fastf_format_umi_rows(C.byref(r), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            _libc_free(out)

    def reset(self):
        check(self._L.fastf_engine_reset(self._h))

    def reseed(self, seed, skip=0):
        check(self._L.fastf_engine_reseed(self._h, seed, skip))

    # ---- timing of individual kernels (bench roofline leg) ----
    def set_timing(self, on: bool):
        check(self._L.fastf_engine_set_timing(self._h, 1 if on else 0))

    def device_records(self, n_devices=1):
        """records each device of a multi-device engine has been given since the last reset"""
        out = (C.c_uint64 * max(1, n_devices))()
        check(self._L.fastf_engine_device_records(self._h, out, max(1, n_devices)))
        return [int(x) for x in out]

    def get_timing(self, which: int):
        ms, n = C.c_double(), C.c_uint64()
        check(self._L.fastf_engine_get_timing(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- device-level API: arguments are raw device pointers (ints) ----
    def reserve(self, max_records, max_keys):
        check(self._L.fastf_dev_reserve(self._h, max_records, max_keys))

    def dev_count_hits(self, d_cb, n, d_out, stream=0):
        check(self._L.fastf_dev_count_hits(self._h, d_cb, n, d_out, stream))

    def dev_count_hits_blocked(self, d_cb, n, d_blocked, d_out, stream=0):
        check(self._L.fastf_dev_count_hits_blocked(self._h, d_cb, n, d_blocked, d_out, stream))

    def block_bytes(self, n) -> int:
        """bytes of the blocked record buffer (gx | umi | meta | cell scratch per 256-record unit) for n records;
        0 = this engine cannot run the streaming K1b that reads it"""
        v = C.c_uint64()
        check(self._L.fastf_dev_block_bytes(self._h, n, C.byref(v)))
        return int(v.value)

    def dev_block_records(self, d_gx, d_umi, d_meta, n, d_blocked, stream=0):
        check(self._L.fastf_dev_block_records(self._h, d_gx, d_umi, d_meta, n, d_blocked, stream))

    def dev_probe_pack(self, d_cb, d_gx, d_umi, d_meta, n, d_draws, n_draws, d_keys, shard_stride,
                       d_key_counts, d_counters, stream=0, d_draw_base=None, reuse_hits=False, segmented=False, blocked=False,
                       draw_bits=False):
        """reuse_hits: dev_count_hits ran on these very records just before, on the same stream (K1a is skipped);
        segmented: streaming K1b, keys land in per-workgroup regions (probe_capacity slots; sort with segmented=True);
        blocked: d_gx is a blocked record buffer (block_bytes / dev_block_records), d_umi and d_meta are ignored;
        draw_bits: d_draws holds the decisions of n_draws draws (dev_draw_bits), not the 32-bit draws themselves"""
        check(self._L.fastf_dev_probe_pack(self._h, d_cb, d_gx, d_umi, d_meta, n, d_draws, n_draws, d_draw_base,
                                           d_keys, shard_stride, d_key_counts, d_counters,
                                           (1 if reuse_hits else 0) | (2 if segmented else 0) | (8 if blocked else 0) | (16 if draw_bits else 0),
                                           stream))

    @property
    def wide(self) -> bool:
        """keys wider than 64 bits (or UMIs beyond 16 bases): the group word and the rest of the key travel in two words"""
        return bool(self._L.fastf_engine_is_wide(self._h))

    def dev_probe_pack_wide(self, d_cb, d_gx, d_umi, d_meta, d_umi_ext, n, d_draws, n_draws, d_keys, d_vals, shard_stride,
                            d_key_counts, d_counters, stream=0, d_draw_base=None, reuse_hits=False, draw_bits=False):
        """dev_probe_pack for an engine whose keys are wider than 64 bits: group words into d_keys[shard][..], the rest of each key
        into d_vals[shard][..]; d_umi_ext: bases 17.. of the UMIs (0 / None: none)"""
        check(self._L.fastf_dev_probe_pack_wide(self._h, d_cb, d_gx, d_umi, d_meta, d_umi_ext or None, n, d_draws, n_draws, d_draw_base,
                                                d_keys, d_vals, shard_stride, d_key_counts, d_counters,
                                                (1 if reuse_hits else 0) | (16 if draw_bits else 0), stream))

    def dev_adopt_wide(self, d_keys, d_vals, n, stream=0):
        """the n (group word, rest of key) pairs this shard owns into the engine's own store: finish() / umi_rows() follow"""
        check(self._L.fastf_dev_adopt_wide(self._h, d_keys, d_vals, n, stream))

    def dev_draw_bits(self, d_draws, n_draws, d_bits_out, stream=0):
        """bit i of d_bits_out = d_draws[i] < this engine's keep threshold.  The kernel stores whole 64-bit words: d_bits_out must
        be 8-byte aligned and hold ((n_draws + 63) // 64) * 8 bytes = ((n_draws + 63) // 64) * 2 u32 words (include/fastf_amd.h)"""
        check(self._L.fastf_dev_draw_bits(self._h, d_draws, n_draws, d_bits_out, stream))

    def dev_mt_decisions(self, seed, skip, n_draws, d_bits_out, stream=0):
        """the decisions of n_draws draws of init_genrand(seed) + skip from the device's own generator (parallel for large
        counts: jump-ahead), bit i of d_bits_out; same size rule as dev_draw_bits"""
        check(self._L.fastf_dev_mt_decisions(self._h, seed, skip, n_draws, d_bits_out, stream))

    def probe_capacity(self, n) -> int:
        """key slots a segmented probe_pack over n records needs; 0 = the streaming form is not available"""
        v = C.c_uint64()
        check(self._L.fastf_dev_probe_capacity(self._h, n, C.byref(v)))
        return int(v.value)

    def dev_set_regions(self, d_counts, n_regions, stride, d_n_out, stream=0):
        """the next segmented dev_sort reads n_regions rows of `stride` key slots, row r holding d_counts[r] keys"""
        check(self._L.fastf_dev_set_regions(self._h, d_counts, n_regions, stride, d_n_out, stream))

    def dev_sort(self, d_keys, d_tmp, d_n, max_n, key_bits=None, stream=0, skip_low=False, segmented=False) -> bool:
        in_tmp = C.c_int(0)
        check(self._L.fastf_dev_sort(self._h, d_keys, d_tmp, d_n, max_n,
                                     self.key_bits if key_bits is None else key_bits,
                                     (2 if skip_low else 0) | (4 if segmented else 0), C.byref(in_tmp), stream))
        return bool(in_tmp.value)

    def dev_reduce(self, d_sorted, d_n, max_n, d_feature, d_cell, d_count, d_nnz, stream=0, skip_low=False, segmented=False):
        """segmented: the rows stay in the engine's row regions (only *d_nnz is written); dev_rows_gather concatenates them"""
        check(self._L.fastf_dev_reduce(self._h, d_sorted, d_n, max_n, d_feature, d_cell, d_count, d_nnz,
                                       (2 if skip_low else 0) | (8 if segmented else 0), stream))

    def dev_rows_gather(self, d_n, feature, cell, count, stream=0):
        """concatenate the row regions of the last dev_reduce into feature/cell/count (device or pinned host pointers)"""
        check(self._L.fastf_dev_rows_gather(self._h, d_n, feature, cell, count, stream))

    @property
    def table_modes(self) -> str:
        c, g = C.c_int(), C.c_int()
        check(self._L.fastf_engine_table_modes(self._h, C.byref(c), C.byref(g)))
        return "cells:%s genes:%s" % ("LDS perfect hash" if c.value else "L2 open addressing",
                                      {0: "L2 open addressing", 1: "LDS bitmap+rank", 2: "LDS direct table"}[g.value])

    @property
    def cell_scratch_bytes(self) -> int:
        b = C.c_uint32()
        check(self._L.fastf_engine_cell_scratch_bytes(self._h, C.byref(b)))
        return b.value

    @property
    def skip_bits(self) -> int:
        b = C.c_uint32()
        check(self._L.fastf_engine_skip_bits(self._h, C.byref(b)))
        return b.value

    def sort_passes(self, skip_low=False) -> int:
        b = C.c_uint32()
        check(self._L.fastf_engine_sort_passes(self._h, 2 if skip_low else 0, C.byref(b)))
        return b.value

    def dev_umi_rows(self, d_sorted, d_n, max_n, d_ukeys, d_ncopy, d_nrows, stream=0):
        check(self._L.fastf_dev_umi_rows(self._h, d_sorted, d_n, max_n, d_ukeys, d_ncopy, d_nrows, stream))

    def dev_error_bits(self) -> int:
        b = C.c_uint64()
        check(self._L.fastf_dev_error_bits(self._h, C.byref(b)))
        return int(b.value)

    def dev_clear_error_bits(self, mask, stream=0):
        check(self._L.fastf_dev_clear_error_bits(self._h, mask, stream))

    def close(self):
        if getattr(self, "_h", None):
            self._L.fastf_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _libc_free(ptr):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(ptr)
