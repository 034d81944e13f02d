"""ctypes binding of libfastf_amd.so (the C ABI declared in include/fastf_amd.h)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB = os.environ.get("FASTF_LIB_OVERRIDE") or os.path.join(HERE, "lib", "libfastf_amd.so")   # override: A/B builds only
_CLI = os.path.join(HERE, "bin", "fastF")


class FastfError(RuntimeError):
    pass


def lib_path():
    return _LIB


def cli_path():
    return _CLI


def build(force=False):
    """Compile every HIP/C source for gfx950 into fastf_amd/lib + fastf_amd/bin (in-tree)."""
    args = ["make", "-s", "-C", os.path.join(HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args + ["all"])


class EngineConfig(C.Structure):
    _fields_ = [
        ("cell_keys", C.c_void_p), ("n_cells", C.c_uint32),
        ("feature_keys", C.c_void_p), ("n_features", C.c_uint32),
        ("draw_threshold", C.c_uint64), ("umi_max_bases", C.c_uint32),
        ("mt_seed", C.c_uint32), ("mt_skip", C.c_uint64),
        ("n_shards", C.c_uint32), ("shard_rank", C.c_uint32),
        ("device", C.c_int32), ("batch_records", C.c_uint64), ("key_capacity", C.c_uint64),
        ("n_devices", C.c_uint32), ("devices", C.c_void_p),
    ]


class Batch(C.Structure):
    _fields_ = [("cb_key", C.c_void_p), ("gx_key", C.c_void_p), ("umi", C.c_void_p),
                ("meta", C.c_void_p), ("n", C.c_size_t), ("umi_ext", C.c_void_p)]


class Coo(C.Structure):
    _fields_ = [("feature", C.POINTER(C.c_uint32)), ("cell", C.POINTER(C.c_uint32)),
                ("count", C.POINTER(C.c_uint32)), ("nnz", C.c_size_t)]


class UmiRows(C.Structure):
    _fields_ = [("feature", C.POINTER(C.c_uint32)), ("cell", C.POINTER(C.c_uint32)),
                ("n_copy", C.POINTER(C.c_uint32)), ("umi", C.POINTER(C.c_uint32)),
                ("nonnull", C.POINTER(C.c_uint8)), ("n", C.c_size_t)]


class ListsStruct(C.Structure):
    _fields_ = [
        ("n_lines_barcodes", C.c_size_t), ("n_sampled_target", C.c_size_t), ("n_cells", C.c_size_t),
        ("barcode", C.POINTER(C.c_char_p)), ("cell_key", C.POINTER(C.c_uint64)),
        ("n_features", C.c_size_t),
        ("feat_id", C.POINTER(C.c_char_p)), ("feat_name", C.POINTER(C.c_char_p)),
        ("feat_type", C.POINTER(C.c_char_p)), ("feature_key", C.POINTER(C.c_uint64)),
        ("cell_dict", C.c_void_p), ("feat_dict", C.c_void_p),
        ("mt_skip", C.c_uint64), ("dup_barcodes", C.c_size_t), ("dup_features", C.c_size_t),
    ]


class TagHistResult(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_valid", C.c_uint64), ("n_key1_present", C.c_uint64),
                ("key1", C.POINTER(C.c_uint64)), ("count1", C.POINTER(C.c_uint64)), ("first1", C.POINTER(C.c_uint64)),
                ("n1", C.c_size_t),
                ("pair_k1", C.POINTER(C.c_uint32)), ("pair_key2", C.POINTER(C.c_uint64)),
                ("pair_count", C.POINTER(C.c_uint64)), ("pair_first", C.POINTER(C.c_uint64)), ("n_pairs", C.c_size_t)]


class MT(C.Structure):
    _fields_ = [("s", C.c_uint32 * 624), ("idx", C.c_int)]


_lib = None

# every symbol include/fastf_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "bam2db", "_umi_copies_flag", "cmd_bam2db", "fastf_last_error", "fastf_version",
    "fastf_mt_seed", "fastf_mt_next", "fastf_mt_fill", "fastf_mt_skip",
    "fastf_draw_threshold", "fastf_sample_cells", "fastf_pack_umi", "fastf_pack_umi_long",
    "fastf_keydict_create", "fastf_keydict_destroy", "fastf_keydict_add", "fastf_keydict_pack",
    "fastf_keydict_pack_many",
    "fastf_engine_create", "fastf_engine_destroy", "fastf_engine_push", "fastf_engine_push_draws",
    "fastf_engine_push_pinned", "fastf_engine_wait_input",
    "fastf_pinned_alloc", "fastf_pinned_free", "fastf_pinned_register", "fastf_pinned_unregister", "fastf_debug_live_registrations", "fastf_pick_devices",
    "fastf_engine_finish", "fastf_engine_lend_rows", "fastf_engine_umi_rows", "fastf_engine_reset", "fastf_engine_key_bits",
    "fastf_engine_skip_bits", "fastf_engine_sort_passes", "fastf_engine_table_modes", "fastf_engine_cell_scratch_bytes",
    "fastf_dev_count_hits", "fastf_dev_count_hits_blocked", "fastf_dev_block_bytes", "fastf_dev_block_records", "fastf_dev_set_regions", "fastf_dev_draw_bits", "fastf_dev_mt_decisions", "fastf_dev_probe_pack", "fastf_dev_probe_pack_wide", "fastf_dev_adopt_wide", "fastf_engine_is_wide", "fastf_dev_probe_capacity", "fastf_dev_sort", "fastf_dev_reduce", "fastf_dev_rows_gather", "fastf_engine_device_records",
    "fastf_dev_umi_rows", "fastf_dev_reserve", "fastf_dev_error_bits",
    "fastf_dev_clear_error_bits", "fastf_kernel_names",
    # crb / extract
    "extract_bam", "read_bam", "free_CB_node", "print_CB_node", "cmd_crb", "cmd_extract",
    "fastf_crb_text", "fastf_extract_text", "fastf_keydict_intern", "fastf_keydict_decode",
    "fastf_taghist_create", "fastf_taghist_destroy", "fastf_taghist_push", "fastf_taghist_finish",
]


def _one_hip_runtime():
    """ONE HIP runtime per process.  The torch wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, the soname this
    library needs too): whichever of the two is loaded first decides which runtime the OTHER gets — torch first, and this
    library binds to torch's copy by soname; this library first, and it brings in /opt/rocm's copy, after which `import torch`
    loads its bundled one BESIDE it (torch asks for the file name libamdhip64.so, which no loaded soname matches): two HIP
    runtimes, two HSA runtimes, one GPU address space.  So: if torch is installed but not imported yet, its runtime is loaded
    here first (the file only — torch itself is not imported), and the order of imports stops mattering."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FASTF_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def hip_runtimes_loaded():
    """paths of every libamdhip64 mapped into this process (one, if all is well)"""
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})


def lib():
    """Load the shared library (one HIP runtime per process: _one_hip_runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise FastfError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no Python/CPU fallback for the engine)" % _LIB)
    _one_hip_runtime()
    L = C.CDLL(_LIB, mode=C.RTLD_GLOBAL)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    L.fastf_last_error.restype = C.c_char_p
    L.fastf_version.restype = C.c_char_p
    L.fastf_kernel_names.restype = C.c_char_p
    L.fastf_dev_probe_capacity.argtypes = [vp, u64, C.POINTER(u64)]
    L.fastf_engine_push_pinned.argtypes = [vp, vp]
    L.fastf_engine_wait_input.argtypes = [vp]
    L.fastf_pinned_alloc.argtypes = [sz]
    L.fastf_pinned_alloc.restype = vp
    L.fastf_pinned_free.argtypes = [vp]
    L.fastf_pinned_register.argtypes = [vp, sz]
    L.fastf_pinned_unregister.argtypes = [vp]
    L.fastf_mt_seed.argtypes = [C.POINTER(MT), u32]
    L.fastf_mt_next.argtypes = [C.POINTER(MT)]
    L.fastf_mt_next.restype = u32
    L.fastf_mt_fill.argtypes = [C.POINTER(MT), vp, sz]
    L.fastf_mt_skip.argtypes = [C.POINTER(MT), u64]
    L.fastf_draw_threshold.argtypes = [C.c_float]
    L.fastf_draw_threshold.restype = u64
    L.fastf_sample_cells.argtypes = [sz, C.c_float, C.c_uint, vp, C.POINTER(sz), C.POINTER(u64)]
    L.fastf_pack_umi.argtypes = [C.c_char_p, sz, C.POINTER(u32)]
    L.fastf_pack_umi.restype = u32
    L.fastf_keydict_create.restype = vp
    L.fastf_keydict_destroy.argtypes = [vp]
    L.fastf_keydict_add.argtypes = [vp, C.c_char_p, sz]
    L.fastf_keydict_add.restype = u64
    L.fastf_keydict_pack.argtypes = [vp, C.c_char_p, sz]
    L.fastf_keydict_pack.restype = u64
    L.fastf_keydict_pack_many.argtypes = [vp, vp, sz, sz, vp, vp]
    L.fastf_keydict_pack_many.restype = None
    L.fastf_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(vp)]
    L.fastf_engine_destroy.argtypes = [vp]
    L.fastf_engine_destroy.restype = None
    L.fastf_engine_push.argtypes = [vp, C.POINTER(Batch)]
    L.fastf_engine_push_draws.argtypes = [vp, C.POINTER(Batch), vp, sz]
    L.fastf_engine_finish.argtypes = [vp, C.POINTER(Coo), C.POINTER(u64 * 3)]
    L.fastf_engine_lend_rows.argtypes = [vp, vp, sz]
    L.fastf_engine_umi_rows.argtypes = [vp, C.POINTER(UmiRows)]
    L.fastf_engine_reset.argtypes = [vp]
    L.fastf_engine_device_records.argtypes = [vp, C.POINTER(u64), u32]
    L.fastf_engine_reseed.argtypes = [vp, u32, u64]
    L.fastf_engine_key_bits.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.fastf_engine_set_timing.argtypes = [vp, C.c_int]
    L.fastf_engine_get_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    L.fastf_dev_count_hits.argtypes = [vp, vp, u64, vp, vp]
    L.fastf_dev_count_hits_blocked.argtypes = [vp, vp, u64, vp, vp, vp]
    L.fastf_dev_block_bytes.argtypes = [vp, u64, C.POINTER(u64)]
    L.fastf_dev_block_records.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    L.fastf_dev_set_regions.argtypes = [vp, vp, u32, u64, vp, vp]
    if hasattr(L, "fastf_dev_draw_bits") or not os.environ.get("FASTF_LIB_OVERRIDE"):   # (an A/B build from before the call existed)
        L.fastf_dev_draw_bits.argtypes = [vp, vp, u64, vp, vp]
    if hasattr(L, "fastf_dev_mt_decisions"):
        L.fastf_dev_mt_decisions.argtypes = [vp, u32, u64, u64, vp, vp]
    L.fastf_dev_probe_pack.argtypes = [vp, vp, vp, vp, vp, u64, vp, u64, vp, vp, u64, vp, vp, u32, vp]
    if hasattr(L, "fastf_dev_probe_pack_wide"):
        L.fastf_dev_probe_pack_wide.argtypes = [vp, vp, vp, vp, vp, vp, u64, vp, u64, vp, vp, vp, u64, vp, vp, u32, vp]
        L.fastf_dev_adopt_wide.argtypes = [vp, vp, vp, u64, vp]
        L.fastf_engine_is_wide.argtypes = [vp]
    L.fastf_dev_sort.argtypes = [vp, vp, vp, vp, u64, u32, u32, C.POINTER(C.c_int), vp]
    L.fastf_dev_reduce.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, u32, vp]
    L.fastf_dev_rows_gather.argtypes = [vp, vp, vp, vp, vp, vp]
    L.fastf_engine_skip_bits.argtypes = [vp, C.POINTER(u32)]
    L.fastf_engine_sort_passes.argtypes = [vp, u32, C.POINTER(u32)]
    L.fastf_engine_table_modes.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fastf_dev_umi_rows.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
    L.fastf_dev_reserve.argtypes = [vp, u64, u64]
    L.fastf_dev_error_bits.argtypes = [vp, C.POINTER(u64)]
    L.fastf_dev_clear_error_bits.argtypes = [vp, u64, vp]
    L.fastf_lists_load_mem.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_float, C.c_uint, C.POINTER(ListsStruct)]
    L.fastf_lists_load.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_uint, C.POINTER(ListsStruct)]
    L.fastf_lists_free.argtypes = [C.POINTER(ListsStruct)]
    L.fastf_lists_free.restype = None
    L.fastf_bam_open.argtypes = [C.c_char_p, C.c_int]
    L.fastf_bam_open.restype = vp
    L.fastf_bam_read_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz]
    L.fastf_bam_read_batch.restype = C.c_long
    L.fastf_bam_close.argtypes = [vp]
    L.fastf_bam_close.restype = None
    L.fastf_bam_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fastf_bam_stats.restype = None
    L.fastf_format_matrix.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(u64 * 3), sz, sz,
                                      C.POINTER(Coo), C.POINTER(vp), C.POINTER(sz)]
    L.fastf_format_umi_rows.argtypes = [C.POINTER(UmiRows), C.POINTER(vp), C.POINTER(sz)]
    L.fastf_keydict_intern.argtypes = [vp, C.c_char_p, sz]
    L.fastf_keydict_intern.restype = u64
    L.fastf_keydict_decode.argtypes = [vp, u64, C.c_char_p, sz]
    L.fastf_keydict_decode.restype = C.c_long
    L.fastf_taghist_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.fastf_taghist_destroy.argtypes = [vp]
    L.fastf_taghist_destroy.restype = None
    L.fastf_taghist_push.argtypes = [vp, vp, vp, sz]
    L.fastf_taghist_finish.argtypes = [vp, C.POINTER(TagHistResult)]
    L.fastf_crb_text.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(sz), C.POINTER(u64)]
    L.fastf_extract_text.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(u64), C.POINTER(u64)]
    L.fastf_bam_read_tags.argtypes = [vp, vp, C.c_char_p, C.c_char_p, C.c_int, vp, vp, sz, C.POINTER(u64)]
    L.fastf_bam_read_tags.restype = C.c_long
    L.read_bam.argtypes = [C.c_char_p]
    L.read_bam.restype = vp
    L.print_CB_node.argtypes = [vp, vp]
    L.print_CB_node.restype = None
    L.free_CB_node.argtypes = [vp]
    L.free_CB_node.restype = None
    L.extract_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.extract_bam.restype = None
    L.bam2db.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_uint]
    _lib = L
    return L


def check(rc):
    if rc:
        raise FastfError(lib().fastf_last_error().decode(errors="replace"))
