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
