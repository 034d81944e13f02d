"""The host reader's batch path under AddressSanitizer + UBSan and under ThreadSanitizer (CPU build, the device side stubbed out:
tools/san_reader.c): windows of two sizes, a record longer than the carry-over reserve, the file unmapped window by window,
the spare window buffer released at end of file — clean reports and the same records whatever the thread count."""
import os
import struct
import subprocess

import pytest

from fastf_amd import synth
from helpers import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "fastf_amd", "csrc", f) for f in ("host_io.c", "host_prims.c", "inflate_fast.c", "crc32_fast.c", "deflate_fast.c")]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_reader_batch_path_is_clean_under_sanitizers(tmp_path, san):
    exe = tmp_path / "san_reader"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=" + san, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fastf_amd", "csrc"),
                           os.path.join(ROOT, "tools", "san_reader.c")] + SRC + ["-lz", "-lpthread", "-o", str(exe)])
    case = Case(n=40_000, n_bar=200, n_gene=80, umi_pool=64, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2, p_n_umi=0.02, p_no_ub=0.03)

    def extra(i):
        return b"ZBBC" + struct.pack("<i", 1_500_000) + bytes(1_500_000) if i == 20_000 else b""
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, shape=lambda i: (91, 1, 0, i % 900), extra_aux=extra)
    (tmp_path / "b.tsv").write_bytes(case.bt); (tmp_path / "f.tsv").write_bytes(case.ft)
    sums = set()
    for window in ("131072", "4194304"):
        r = subprocess.run([str(exe), str(bam), str(tmp_path / "b.tsv"), str(tmp_path / "f.tsv")], capture_output=True, text=True,
                           env=dict(os.environ, FASTF_BAM_WINDOW=window))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Sanitizer" not in r.stderr, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("threads")]
        assert len(lines) == 4 and all("40000 records" in l for l in lines)
        sums |= {l.split("checksum")[1].strip() for l in lines}
    assert len(sums) == 1


BAM2DB_SRC = [os.path.join(ROOT, "fastf_amd", "csrc", "bam2db_main.c")] + SRC


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_bam2db_host_side_is_clean_under_sanitizers(tmp_path, san):
    """bam2db()'s threads (decoder and its ring of slots, pin thread, release thread, scout, prefetch, writers and their side thread)
    around a stub engine that comes up late and checks the push order (tools/san_bam2db.c): clean reports, every record pushed
    once, in file order — the same checksum whatever the slot and window sizes, and the reader harness's checksum."""
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fastf_amd", "csrc"), "-I" + os.path.join(ROOT, "tools")]
    exe, rdr = tmp_path / "san_bam2db", tmp_path / "san_reader"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=" + san] + inc + [os.path.join(ROOT, "tools", "san_bam2db.c")] + BAM2DB_SRC + ["-lz", "-lpthread", "-o", str(exe)])
    subprocess.check_call(["gcc", "-O1", "-g"] + inc + [os.path.join(ROOT, "tools", "san_reader.c")] + SRC + ["-lz", "-lpthread", "-o", str(rdr)])
    case = Case(n=150_000, n_bar=500, n_gene=200, umi_pool=512, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2, p_n_umi=0.02)
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    (tmp_path / "b.tsv").write_bytes(case.bt); (tmp_path / "f.tsv").write_bytes(case.ft)
    out = tmp_path / "out"; out.mkdir()
    r = subprocess.run([str(rdr), str(bam), str(tmp_path / "b.tsv"), str(tmp_path / "f.tsv")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    want = r.stdout.splitlines()[0].split("checksum")[1].strip()
    runs = [(dict(FASTF_BATCH_RECORDS="70000", FASTF_BAM_WINDOW="131072"), []),                      # the ring wraps; two windows per slot
            (dict(FASTF_BATCH_RECORDS="4194304", FASTF_BAM_WINDOW="1048576"), ["x"]),                # one big slot filled across windows; leaves like the CLI
            (dict(FASTF_BATCH_RECORDS="20000", FASTF_BAM_WINDOW="131072", FASTF_ZERO_COPY="0", FASTF_LEND_ROWS="0", FASTF_BAM_SCOUT="0"), []),
            (dict(FASTF_BATCH_RECORDS="50000", FASTF_BAM_WINDOW="262144", FASTF_BAM_MMAP="0"), [])]
    for env_extra, mode in runs:
        r = subprocess.run([str(exe), str(bam), str(tmp_path / "b.tsv"), str(tmp_path / "f.tsv"), str(out)] + mode, capture_output=True, text=True,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=%d" % (0 if mode else 1), **env_extra))
        assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, r.stderr[-1000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("bam2db returned")][0]
        assert "returned 0: pushed %d records" % case.n in line and line.split("checksum")[1].strip() == want, line
        assert "total fastQ reads: %d" % case.n in r.stdout
        assert (out / "matrix.mtx.gz").exists() and (out / "features.tsv.gz").exists()


def test_reader_refuses_or_parses_mutated_bams_cleanly(tmp_path):
    """tools/fuzz_reader.py, a short round: damaged BGZF containers are all refused, damaged record payloads (re-wrapped with
    valid CRCs) either parse to the same records on every thread count or end in an error — no sanitizer report, no hang"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_reader", os.path.join(ROOT, "tools", "fuzz_reader.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    findings, outcomes = mod.run(60, seed=3, workdir=str(tmp_path))
    assert not findings, findings[:2]
    assert 0 not in outcomes["container"] and outcomes["records"].get(0, 0) > 10
