"""Mutation fuzz of the BAM reader's batch path under AddressSanitizer + UBSan (CPU, tools/san_reader.c):
  * container level — bit flips, truncation, garbage runs, damaged block headers, dropped / repeated slices of the BGZF file
    (every one must be refused: a CRC, a header field or the framing says so);
  * record level — the same kinds of damage applied to the INFLATED payload, re-wrapped in valid BGZF blocks with correct CRCs,
    so that the record hop and the tag parser see it (whatever parses must parse to the same records on 1, 2, 4 and 8 threads).
  * device record functions — every record-level input also goes through gpu_records.hpp's host build (tools/fuzz_records.cpp):
    the chain walk, the chain-start guess and the tag/key packer over a buffer that ends where the payload ends, compared with
    what the host reader delivered.
A finding is a sanitizer report, a crash, a hang, thread counts that disagree, or a record the two packers disagree on.

    python tools/fuzz_reader.py [iterations per level] [seed]        (tests/test_sanitizers.py runs a short round)
"""
import os
import random
import struct
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
SRC = [os.path.join(ROOT, "fastf_amd", "csrc", f) for f in ("host_io.c", "host_prims.c", "inflate_fast.c", "crc32_fast.c", "deflate_fast.c")]


def build(exe):
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "fastf_amd", "csrc"), "-I" + os.path.join(ROOT, "tools"),
                           os.path.join(ROOT, "tools", "san_reader.c")] + SRC + ["-lz", "-lpthread", "-o", exe])


def build_records(exe):
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fastf_amd", "csrc"), "-I" + os.path.join(ROOT, "tools")]
    objs = []
    for c in SRC + [os.path.join(ROOT, "tools", "san_stubs.c")]:
        o = exe + "_" + os.path.basename(c) + ".o"
        subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined"] + inc + ["-c", c, "-o", o]); objs.append(o)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined"] + inc +
                          [os.path.join(ROOT, "tools", "fuzz_records.cpp"), os.path.join(ROOT, "tools", "gr_host.cpp")] + objs + ["-lz", "-lpthread", "-o", exe])


def bgzf(data, blk=20000):
    out = bytearray()
    for a in list(range(0, len(data), blk)) + [None]:
        chunk = b"" if a is None else bytes(data[a:a + blk])
        c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(chunk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def run(iterations=100, seed=1, workdir=None, log=print):
    from helpers import Case
    from fastf_amd import synth
    d = workdir or tempfile.mkdtemp(prefix="fuzz_reader_")
    exe, exe_rec = os.path.join(d, "san_reader"), os.path.join(d, "fuzz_records")
    build(exe)
    build_records(exe_rec)
    case = Case(n=6000, n_bar=60, n_gene=30, umi_pool=64, p_no_cb=0.05, p_bad_xf=0.2)
    bam, mb = os.path.join(d, "t.bam"), os.path.join(d, "m.bam")
    synth.write_bam(bam, case.flags, case.xf, case.cb, case.gx, case.ub)
    bt, ft = os.path.join(d, "b.tsv"), os.path.join(d, "f.tsv")
    open(bt, "wb").write(case.bt); open(ft, "wb").write(case.ft)
    raw = open(bam, "rb").read()
    pos, payload = 0, bytearray()
    while pos + 18 < len(raw):
        xlen = struct.unpack_from("<H", raw, pos + 10)[0]; bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
        comp = raw[pos + 12 + xlen: pos + bsize - 8]
        payload += zlib.decompress(comp, -15) if len(comp) else b""
        pos += bsize
    l_text = struct.unpack_from("<i", payload, 4)[0]; n_ref = struct.unpack_from("<i", payload, 8 + l_text)[0]
    off = 12 + l_text
    for _ in range(n_ref):
        off += 8 + struct.unpack_from("<i", payload, off)[0]
    rng = random.Random(seed)
    env = dict(os.environ, FASTF_BAM_WINDOW="131072", ASAN_OPTIONS="detect_leaks=0")     # (the harness leaks on its error exits)
    findings, outcomes = [], {"container": {}, "records": {}, "device_functions": {}}
    pb = os.path.join(d, "m.payload")

    def device_functions(it, m):
        open(pb, "wb").write(bytes(m))
        try:
            r = subprocess.run([exe_rec, mb, pb, str(off), bt, ft], capture_output=True, text=True, timeout=120, env=env)
        except subprocess.TimeoutExpired:
            findings.append(("device_functions", it, "hang")); return
        outcomes["device_functions"][r.returncode] = outcomes["device_functions"].get(r.returncode, 0) + 1
        if "Sanitizer" in r.stderr or "runtime error" in r.stderr or r.returncode != 0:
            findings.append(("device_functions", it, r.returncode, r.stdout[-200:], r.stderr[:1500]))

    def one(level, it, blob):
        open(mb, "wb").write(blob)
        try:
            r = subprocess.run([exe, mb, bt, ft], capture_output=True, text=True, timeout=120, env=env)
        except subprocess.TimeoutExpired:
            findings.append((level, it, "hang")); return
        outcomes[level][r.returncode] = outcomes[level].get(r.returncode, 0) + 1
        lines = [l for l in r.stdout.splitlines() if l.startswith("threads")]
        same = len({l.split(":", 1)[1] for l in lines}) <= 1
        if "Sanitizer" in r.stderr or "runtime error" in r.stderr or r.returncode < 0 or r.returncode > 9 or (r.returncode == 0 and not same) or \
           (level == "container" and r.returncode == 0):
            findings.append((level, it, r.returncode, r.stderr[:1500]))

    for it in range(iterations):
        m = bytearray(raw); kind = it % 5
        if kind == 0:
            for _ in range(rng.randint(1, 4)): m[rng.randrange(len(m) - 28)] ^= 1 << rng.randrange(8)      # (not the empty EOF block: a flip there may be harmless)
        elif kind == 1: m = m[:rng.randrange(20, len(m) - 28)]
        elif kind == 2:
            a = rng.randrange(len(m) - 100); n = rng.randint(1, 64); m[a:a + n] = bytes(rng.randrange(256) for _ in range(n))
            if bytes(m) == raw: continue
        elif kind == 3:
            a = rng.randrange(0, len(m) - 60); b0 = a + rng.randrange(18); m[b0] = (m[b0] + 1 + rng.randrange(255)) & 255
        else:
            a = rng.randrange(len(m) - 300); n = rng.randint(1, 200)
            m = m[:a] + m[a:a + n] + m[a:] if it % 2 else m[:a] + m[a + n:]
        one("container", it, bytes(m))
    for it in range(iterations):
        m = bytearray(payload); kind = it % 4
        if kind == 0:
            for _ in range(rng.randint(1, 6)): m[rng.randrange(off, len(m))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            a = rng.randrange(off, len(m) - 4)
            struct.pack_into("<I", m, a, rng.choice([0, 1, 31, 32, 0x7fffffff, 0xffffffff, rng.randrange(1 << 32)]))
        elif kind == 2: m = m[:rng.randrange(off, len(m))]
        else:
            a = rng.randrange(off, len(m) - 64); n = rng.randint(1, 64); m[a:a + n] = bytes(rng.randrange(256) for _ in range(n))
        one("records", it, bgzf(m))
        device_functions(it, m)
    log("fuzz_reader: %d + %d inputs, return codes %s, findings %d" % (iterations, iterations, outcomes, len(findings)))
    return findings, outcomes


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for x in f: print(x)
    sys.exit(1 if f else 0)
