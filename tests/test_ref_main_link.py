"""Drop-in link test: the reference's OWN main.c / argparse.c / filter.c /
count.c, compiled in place from /root/reference and linked against libfastf_amd.so instead of bam2db_ds.c, extract.c,
hashtable.c, mt19937ar.c, utils.c, htslib and sqlite3 (`make -C oracle refcli`).  The resulting binary binds `bam2db`,
`_umi_copies_flag`, `read_bam`, `extract_bam`, `print_CB_node`, `free_CB_node` from the library, and its command-line
handling — the reference's argparse.c — pins the product CLI's own parser (cmd_bam2db, main.c:288-362).
The binary is test infrastructure (oracle/_ref/, git-ignored); nothing of the product links or runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFMAIN = os.path.join(ROOT, "oracle", "_ref", "fastF_refmain")
OURS = os.path.join(ROOT, "fastf_amd", "bin", "fastF")


@pytest.fixture(scope="module")
def refmain():
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refcli"], stderr=subprocess.DEVNULL)
    if not os.path.exists(REFMAIN):
        pytest.skip("reference sources not present and no prebuilt oracle/_ref/fastF_refmain")
    return REFMAIN


def test_reference_main_binds_the_drop_in_symbols(refmain):
    und = subprocess.run(["nm", "-D", "--undefined-only", refmain], capture_output=True, text=True, check=True).stdout
    for sym in ("bam2db", "read_bam", "extract_bam", "print_CB_node", "free_CB_node"):
        assert (" U " + sym + "\n") in und, sym
    alls = subprocess.run(["nm", "-D", refmain], capture_output=True, text=True, check=True).stdout
    assert "_umi_copies_flag" in alls                     # data symbol: bound through a copy relocation
    ldd = subprocess.run(["ldd", refmain], capture_output=True, text=True, check=True).stdout
    assert "libfastf_amd.so" in ldd and "not found" not in ldd
    assert "libhts" not in ldd and "sqlite" not in ldd
    # every undefined symbol that is not libc / libz / libgomp / libm resolves into the product library
    lib = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "fastf_amd", "lib", "libfastf_amd.so")],
                         capture_output=True, text=True, check=True).stdout
    for sym in ("bam2db", "_umi_copies_flag", "read_bam", "extract_bam", "print_CB_node", "free_CB_node"):
        assert (" " + sym + "\n") in lib, sym


def test_reference_main_runs_with_the_library(refmain):
    p = subprocess.run([refmain, "bam2db", "-h"], capture_output=True, text=True)
    assert p.returncode == 0 and "--umicopies" in p.stdout and "--barcode" in p.stdout
    p = subprocess.run([refmain, "crb", "-h"], capture_output=True, text=True)
    assert p.returncode == 0
    p = subprocess.run([refmain, "extract", "-h"], capture_output=True, text=True)
    assert p.returncode == 0


BAD_ARGS = [
    ["-x"], ["--bogus=1"], ["-c"], ["--cell"], ["-c", "abc", "-b", "{bam}"], ["--cell=0.5x"], ["-c0.5x"], ["-s", "12x"], ["--seed=z"],
    ["-r", ""], ["-c", "1e999"], ["-s", "99999999999999999999"], ["-b", "/nonexistent", "-f", "{tsv}", "-a", "{tsv}"], ["-b", "{bam}", "-f", "/nonexistent", "-a", "{tsv}"],
    ["-b", "{bam}", "-f", "{tsv}", "-a", "/nonexistent"], ["-b", "{bam}", "-f", "{tsv}", "-a", "{tsv}", "-d", "{tsv}"],
    ["-s", "0x39e", "-b", "/nonexistent"], ["-c0.5", "-b", "/nonexistent"], ["--seed", "7", "--cell", "0.25", "--bam", "/nonexistent"],
    ["--bam=/nonexistent", "-r", "1e-1"],
]


@pytest.mark.parametrize("args", BAD_ARGS, ids=[" ".join(a) or "-" for a in BAD_ARGS])
def test_cmd_bam2db_parsing_matches_the_reference_cli(refmain, tmp_path, args):
    """same exit status and the same first line on stderr as the reference's main.c + argparse.c, for option errors
    (argparse.c:36-46, 88-108, 149-197, 274-277) and for the pre-checks of main.c:323-345"""
    bam = tmp_path / "e.bam"; bam.write_bytes(b"")
    tsv = tmp_path / "e.tsv"; tsv.write_bytes(b"")
    argv = [a.format(bam=bam, tsv=tsv) for a in args]
    ref = subprocess.run([refmain, "bam2db"] + argv, capture_output=True, text=True)
    our = subprocess.run([OURS, "bam2db"] + argv, capture_output=True, text=True)
    assert our.returncode == ref.returncode
    assert our.stderr.splitlines()[:1] == ref.stderr.splitlines()[:1]
