"""The C-ABI library loads without a GPU, exports every symbol include/fastf_amd.h declares,
and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fastf_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(fastf_[a-z0-9_]+|bam2db|cmd_bam2db|cmd_crb|cmd_extract|extract_bam|read_bam|free_CB_node|"
                           r"_umi_copies_flag)\s*\(", txt))
    names.add("_umi_copies_flag")
    names.add("print_CB_node")          # declared in host_io.h / the reference's extract.h (zlib's gzFile in its signature)
    names -= {"fastf_mt", "fastf_engine_config", "fastf_batch", "fastf_coo", "fastf_umi_rows", "fastf_taghist_result"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.lib_path())
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    for name in _lib.ABI_SYMBOLS:
        assert name in declared or name == "_umi_copies_flag", name


def test_no_reference_or_oracle_in_the_product_binary():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    ldd = subprocess.run(["ldd", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "sqlite" not in ldd


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_engine_fails_loudly_without_gpu():
    cells = np.array([1 << 62 | 5], dtype=np.uint64)
    feats = np.array([2 << 62 | 7], dtype=np.uint64)
    with pytest.raises(F.FastfError) as ei:
        F.Engine(cells, feats)
    assert "no CPU fallback" in str(ei.value)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_cli_fails_loudly_without_gpu(tmp_path):
    from fastf_amd import synth
    bt, ft, bar, genes = synth.make_lists(5, 3)
    fl, xf, cb, gx, ub = synth.make_records(20, bar, genes)
    bam = tmp_path / "x.bam"
    synth.write_bam(str(bam), fl, xf, cb, gx, ub)
    (tmp_path / "b.tsv").write_bytes(bt)
    (tmp_path / "f.tsv").write_bytes(ft)
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(tmp_path / "b.tsv"), "-f", str(tmp_path / "f.tsv"),
                        "-o", str(tmp_path), "-c", "1", "-r", "1"], capture_output=True, text=True)
    assert r.returncode == 1
    assert "bam2db failed" in r.stderr and "no CPU fallback" in r.stderr
    assert not (tmp_path / "matrix.mtx.gz").exists()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_tag_commands_fail_loudly_without_gpu(tmp_path):
    from fastf_amd.tags import TagHist, crb_text
    with pytest.raises(F.FastfError) as ei:
        TagHist()
    assert "no CPU fallback" in str(ei.value)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tag_helpers import TagCase
    c = TagCase(n=50, n_cb=3)
    c.write(str(tmp_path / "t.bam"))
    with pytest.raises(F.FastfError):
        crb_text(str(tmp_path / "t.bam"))
    r = subprocess.run([_lib.cli_path(), "crb", "-b", str(tmp_path / "t.bam"), "-o", str(tmp_path / "o.gz")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = subprocess.run([_lib.cli_path(), "extract", "-b", str(tmp_path / "t.bam"), "-t", "CB"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not (tmp_path / "tag_summary.csv").exists()


def test_cli_argument_errors(tmp_path):
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(tmp_path / "missing.bam")], capture_output=True, text=True)
    assert r.returncode == 1 and "does not exist" in r.stderr
    r = subprocess.run([_lib.cli_path(), "bam2db", "--bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown option" in r.stderr
    r = subprocess.run([_lib.cli_path(), "bam2db", "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--umicopies" in r.stdout


def test_cpp_exceptions_do_not_cross_the_c_abi():
    """bam2db_ds.h:60: failure = return 1 + a message.  An exception thrown by host-side C++ (a row buffer sized from a
    device counter, a per-device list) must come back as that, not as std::terminate -> SIGABRT in the caller."""
    L = C.CDLL(_lib.lib_path())
    L.fastf_debug_raise_.restype = C.c_int
    L.fastf_last_error.restype = C.c_char_p
    expect = {0: "vector", 1: "out of host memory", 2: "unknown C++ exception", 3: "vector"}
    for kind, text in expect.items():
        assert L.fastf_debug_raise_(kind) == 1
        msg = L.fastf_last_error().decode()
        assert msg.startswith("fastf_debug_raise_: ") and text in msg, msg
    assert L.fastf_debug_raise_(99) == 0


def test_every_extern_c_body_in_the_cpp_unit_is_behind_the_barrier():
    """static check: an extern "C" definition of the C++ translation unit either fits one line (getter / setter, no
    allocation) or is a function-try-block closed by FASTF_CATCH_*"""
    csrc = os.path.join(ROOT, "fastf_amd", "csrc")
    n = 0
    for name in ("umi_engine.hip", "multi_engine.hpp", "tag_hist.hpp", "gpu_frontend.hpp", "gpu_records.hpp"):
        path = os.path.join(csrc, name)
        if not os.path.exists(path):
            continue
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if not re.match(r'extern "C" .*\bfastf_\w+\(', line) or line.rstrip().endswith("}"):
                continue
            j = i
            while not lines[j].rstrip().endswith("{"):
                j += 1
            if "fastf_last_error" in line:
                continue
            assert lines[j].rstrip().endswith("FASTF_TRY {"), "%s:%d has no exception barrier" % (name, i + 1)
            k = j + 1
            while not lines[k].startswith("}"):
                k += 1
            assert lines[k].startswith("} FASTF_CATCH_"), "%s:%d" % (name, k + 1)
            n += 1
    assert n >= 40


def test_no_result_changing_build_knob_in_the_product():
    """knobs that change results are experiment-only: named FASTF_X_*, compiled only under FASTF_EXPERIMENT, which `make all`
    never defines and which the library would announce in its version string"""
    csrc = os.path.join(ROOT, "fastf_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "FASTF_EXPERIMENT" not in mk and "FASTF_X_" not in mk
    L = C.CDLL(_lib.lib_path())
    L.fastf_version.restype = C.c_char_p
    assert b"EXPERIMENT" not in L.fastf_version()
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hpp", ".hip", ".c", ".h")):
            continue
        for i, line in enumerate(open(os.path.join(csrc, fn), errors="replace"), 1):
            if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", line) and "FASTF_X_" in line and "FASTF_X_ANY" not in line:
                assert "FASTF_EXPERIMENT" in line, "%s:%d: FASTF_X_ knob outside FASTF_EXPERIMENT" % (fn, i)
            # the names round 3's verdict listed must not come back under another guard
            assert not re.search(r"FASTF_K3_(NOHASH|HASH_NOFLAG|DEBUG)\b", line) or line.lstrip().startswith("//"), "%s:%d" % (fn, i)


def test_no_pageable_transfers_in_the_python_layer():
    """DESIGN section 14, rule 1: no code of the repo hands pageable host memory to the HIP runtime — torch's `.cpu()` and
    `torch.from_numpy(x).to(device)` / `.cuda()` of arrays are spelled fastf_amd.hostmem.to_host / to_device (pinned staging)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py"), os.path.join(root, "tools", "config5_whole.py")]
    for d in ("fastf_amd", "tests"):
        files += [os.path.join(root, d, f) for f in os.listdir(os.path.join(root, d)) if f.endswith(".py")]
    bad = []
    for f in files:
        if f.endswith(os.path.join("fastf_amd", "hostmem.py")) or f.endswith("test_abi.py"):
            continue
        for i, ln in enumerate(open(f), 1):
            if re.search(r"\.cpu\(\)|from_numpy\([^)]*\)\s*\.(to|cuda)\(", ln):
                bad.append("%s:%d: %s" % (os.path.relpath(f, root), i, ln.strip()))
    assert not bad, "\n".join(bad)
