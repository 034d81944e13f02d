"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  It loads ``oracle/libfastf_oracle.so`` (the C restatement of
the reference bam2db path, see fastf_oracle.h) and, when present,
``oracle/_ref/libfastf_ref.so`` (the reference's own libc-only sources compiled in
place by ``make -C oracle ref``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfastf_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libfastf_ref.so")
REF_TREE_PATH = os.path.join(HERE, "_ref", "libfastf_ref_tree.so")

HAS_CB, HAS_XF, HAS_GX, HAS_UB = 1, 2, 4, 8
MAX_BLOB = 16


class _Result(C.Structure):
    _fields_ = [
        ("total_reads", C.c_uint64), ("sampled_reads", C.c_uint64),
        ("sampled_valid_reads", C.c_uint64), ("undefined_records", C.c_uint64),
        ("n_feature", C.c_size_t), ("n_barcode", C.c_size_t), ("nnz", C.c_size_t),
        ("mtx_feature", C.POINTER(C.c_int32)), ("mtx_cell", C.POINTER(C.c_int32)),
        ("mtx_count", C.POINTER(C.c_int32)),
        ("n_umi_rows", C.c_size_t),
        ("umi_feature", C.POINTER(C.c_int32)), ("umi_cell", C.POINTER(C.c_int32)),
        ("umi_ncopy", C.POINTER(C.c_int32)), ("umi_text", C.POINTER(C.c_char)),
        ("matrix_txt", C.POINTER(C.c_char)), ("matrix_len", C.c_size_t),
        ("barcodes_txt", C.POINTER(C.c_char)), ("barcodes_len", C.c_size_t),
        ("features_txt", C.POINTER(C.c_char)), ("features_len", C.c_size_t),
        ("umi_txt", C.POINTER(C.c_char)), ("umi_len", C.c_size_t),
        ("n_rows", C.c_size_t),
        ("row_cell", C.POINTER(C.c_int32)), ("row_feature", C.POINTER(C.c_int32)),
        ("row_blob_len", C.POINTER(C.c_int16)), ("row_blob", C.POINTER(C.c_uint8)),
        ("n_sampled", C.c_size_t), ("sampled_lines", C.POINTER(C.c_uint64)),
        ("err", C.c_char * 256),
    ]


_lib = None


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    srcs = [os.path.join(HERE, f) for f in ("fastf_oracle.c", "fastf_oracle_tags.c", "fastf_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle"])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_PATH) or not os.path.exists(REF_TREE_PATH)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_bam2db.restype = C.c_int
        L.oracle_bam2db.argtypes = [
            C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
            C.c_size_t, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
            C.c_float, C.c_float, C.c_uint, C.c_char_p, C.c_int, C.POINTER(_Result)]
        L.oracle_result_free.argtypes = [C.POINTER(_Result)]
        L.oracle_init_genrand.argtypes = [C.c_uint32]
        L.oracle_genrand_int32.restype = C.c_uint32
        L.oracle_genrand_real1.restype = C.c_double
        L.oracle_sample_cells.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p]
        L.oracle_sample_cells.restype = C.c_int
        L.oracle_n_cells_sampled.argtypes = [C.c_size_t, C.c_float]
        L.oracle_n_cells_sampled.restype = C.c_size_t
        L.oracle_djb2.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_djb2.restype = C.c_uint64
        L.oracle_encode_dna.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.oracle_encode_dna.restype = C.c_int
        L.oracle_decode_dna.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
        L.oracle_keep_draw.argtypes = [C.c_uint32, C.c_float]
        L.oracle_keep_draw.restype = C.c_int
        L.oracle_extract.restype = C.c_int
        L.oracle_extract.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oracle_crb.restype = C.c_int
        L.oracle_crb.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oracle_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def run_extract(present, vals=None, ivals=None):
    """extract_bam restated: vals = numpy 'S<k>' array (type 0) or ivals = int64 array (type 1).
    Returns (csv bytes, total_printed, valid)."""
    L = lib()
    present = np.ascontiguousarray(present, dtype=np.uint8)
    n = len(present)
    out, ln, tot, val = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
    if ivals is None:
        v, st = _fixed(vals)
        L.oracle_extract(n, present.ctypes.data, v.ctypes.data, st, None, 0, C.byref(out), C.byref(ln), C.byref(tot), C.byref(val))
    else:
        iv = np.ascontiguousarray(ivals, dtype=np.int64)
        L.oracle_extract(n, present.ctypes.data, None, 0, iv.ctypes.data, 1, C.byref(out), C.byref(ln), C.byref(tot), C.byref(val))
    txt = C.string_at(out.value, ln.value)
    L.oracle_free(out)
    return txt, int(tot.value), int(val.value)


def run_crb(has_cb, has_cr, cb, cr):
    """read_bam + print_CB_node restated.  Returns (text bytes, read_count, undefined)."""
    L = lib()
    has_cb = np.ascontiguousarray(has_cb, dtype=np.uint8)
    has_cr = np.ascontiguousarray(has_cr, dtype=np.uint8)
    cb, cbs = _fixed(cb)
    cr, crs = _fixed(cr)
    out, ln, rc_, ud = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
    L.oracle_crb(len(has_cb), has_cb.ctypes.data, has_cr.ctypes.data, cb.ctypes.data, cbs, cr.ctypes.data, crs,
                 C.byref(out), C.byref(ln), C.byref(rc_), C.byref(ud))
    txt = C.string_at(out.value, ln.value)
    L.oracle_free(out)
    return txt, int(rc_.value), int(ud.value)


def ref_tree_lib():
    """The reference's own filter.c (insert_tree / print_tree / print_tree_same_row), None if not built."""
    if not os.path.exists(REF_TREE_PATH):
        return None
    R = C.CDLL(REF_TREE_PATH)
    R.insert_tree.argtypes = [C.c_void_p, C.c_char_p]
    R.insert_tree.restype = C.c_void_p
    R.print_tree.argtypes = [C.c_void_p, C.c_void_p]
    R.print_tree_same_row.argtypes = [C.c_void_p, C.c_void_p]
    R.free_tree_node.argtypes = [C.c_void_p]
    return R


def ref_lib():
    """The reference's own mt19937ar.c/utils.c/hashtable.c (None if not built)."""
    if not os.path.exists(REF_PATH):
        return None
    R = C.CDLL(REF_PATH)
    R.init_genrand.argtypes = [C.c_ulong]
    R.genrand_int32.restype = C.c_ulong
    R.genrand_real1.restype = C.c_double
    R.GetSeqInt.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
    R.GetSeqInt.restype = C.POINTER(C.c_size_t)
    R.SampleInt.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_uint, C.c_uint]
    R.SampleInt.restype = C.POINTER(C.c_size_t)
    R.hash_table_create.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    R.hash_table_create.restype = C.c_void_p
    R.hash_table_insert.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    R.hash_table_insert.restype = C.c_bool
    R.hash_table_lookup.argtypes = [C.c_void_p, C.c_char_p]
    R.hash_table_lookup.restype = C.c_void_p
    return R


def mt_stream(seed, n, skip=0, state=False):
    """n draws of genrand_int32() after init_genrand(seed) and `skip` draws thrown away (oracle_mt_draws: the loop in C);
    state=True: also the generator's 624-word array afterwards"""
    L = lib()
    L.oracle_mt_draws.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.oracle_mt_draws.restype = None
    out = np.zeros(max(int(n), 1), dtype=np.uint32)
    st = np.zeros(624, dtype=np.uint32)
    L.oracle_mt_draws(seed, int(skip), int(n), out.ctypes.data, st.ctypes.data)
    return (out[:n], st) if state else out[:n]


def sample_cells(n_total, n_sample, seed):
    out = np.zeros(max(n_total, 1), dtype=np.uint64)
    rc = lib().oracle_sample_cells(n_total, n_sample, seed, out.ctypes.data)
    if rc:
        raise ValueError("sample size larger than population")
    return out[:n_sample].copy()


def encode_dna(seq: bytes):
    buf = np.zeros(MAX_BLOB, dtype=np.uint8)
    nb = C.c_size_t(0)
    rc = lib().oracle_encode_dna(seq, buf.ctypes.data, MAX_BLOB, C.byref(nb))
    if rc == -1:
        return None
    if rc:
        raise ValueError("UMI too long for oracle")
    return bytes(buf[:nb.value])


def _fixed(arr):
    a = np.ascontiguousarray(arr)
    assert a.dtype.kind == "S"
    return a, a.dtype.itemsize


def run_bam2db(barcodes: bytes, features: bytes, flags, xf, cb, gx, ub,
               rate_cell=1.0, rate_depth=1.0, seed=926, bam_label=b"synthetic.bam",
               umi_copies=False, want_rows=False):
    """Run the oracle.  cb/gx/ub are numpy 'S<k>' arrays whose itemsize leaves room for
    the terminating NUL (use records.as_cstr()).  Returns a dict of numpy arrays/bytes."""
    L = lib()
    flags = np.ascontiguousarray(flags, dtype=np.uint8)
    xf = np.ascontiguousarray(xf, dtype=np.int32)
    cb, cbs = _fixed(cb)
    gx, gxs = _fixed(gx)
    ub, ubs = _fixed(ub)
    n = len(flags)
    assert len(xf) == n and len(cb) == n and len(gx) == n and len(ub) == n
    res = _Result()
    rc = L.oracle_bam2db(barcodes, len(barcodes), features, len(features), n,
                         flags.ctypes.data, xf.ctypes.data,
                         cb.ctypes.data, cbs, gx.ctypes.data, gxs, ub.ctypes.data, ubs,
                         C.c_float(rate_cell), C.c_float(rate_depth), seed, bam_label,
                         1 if umi_copies else 0, C.byref(res))
    try:
        if rc:
            raise RuntimeError("oracle: " + res.err.decode())

        def arr(ptr, n_, dt):
            if n_ == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n_,)).astype(dt, copy=True)

        out = dict(
            total=int(res.total_reads), sampled=int(res.sampled_reads),
            valid=int(res.sampled_valid_reads), undefined=int(res.undefined_records),
            n_feature=int(res.n_feature), n_barcode=int(res.n_barcode), nnz=int(res.nnz),
            feature=arr(res.mtx_feature, res.nnz, np.int32),
            cell=arr(res.mtx_cell, res.nnz, np.int32),
            count=arr(res.mtx_count, res.nnz, np.int32),
            matrix=C.string_at(res.matrix_txt, res.matrix_len),
            barcodes=C.string_at(res.barcodes_txt, res.barcodes_len),
            features=C.string_at(res.features_txt, res.features_len),
            umi=C.string_at(res.umi_txt, res.umi_len) if umi_copies else b"",
            sampled_lines=arr(res.sampled_lines, res.n_sampled, np.uint64),
            n_umi_rows=int(res.n_umi_rows),
            umi_feature=arr(res.umi_feature, res.n_umi_rows, np.int32),
            umi_cell=arr(res.umi_cell, res.n_umi_rows, np.int32),
            umi_ncopy=arr(res.umi_ncopy, res.n_umi_rows, np.int32),
        )
        if res.n_umi_rows:
            t = np.ctypeslib.as_array(C.cast(res.umi_text, C.POINTER(C.c_uint8)),
                                      shape=(res.n_umi_rows, 11)).copy()
            out["umi_text"] = [bytes(r).split(b"\0")[0] for r in t]
        else:
            out["umi_text"] = []
        if want_rows:
            out["row_cell"] = arr(res.row_cell, res.n_rows, np.int32)
            out["row_feature"] = arr(res.row_feature, res.n_rows, np.int32)
            out["row_blob_len"] = arr(res.row_blob_len, res.n_rows, np.int16)
            if res.n_rows:
                out["row_blob"] = np.ctypeslib.as_array(
                    res.row_blob, shape=(res.n_rows, MAX_BLOB)).copy()
            else:
                out["row_blob"] = np.zeros((0, MAX_BLOB), dtype=np.uint8)
        return out
    finally:
        L.oracle_result_free(C.byref(res))
