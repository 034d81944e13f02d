"""Seeded synthetic inputs for the bam2db path (SURVEY.md §8d workloads) and a minimal
BGZF/BAM writer for end-to-end tests.  Generators use numpy's PCG64 — never the MT
stream under test."""
import struct
import zlib

import numpy as np

HAS_CB, HAS_XF, HAS_GX, HAS_UB = 1, 2, 4, 8
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _kmers(codes_u64: np.ndarray, k: int) -> np.ndarray:
    """integer codes → (n, k) uint8 ASCII bases, first base from the top bits"""
    shifts = (2 * (k - 1 - np.arange(k))).astype(np.uint64)
    idx = ((codes_u64[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.intp)
    return _BASES[idx]


def _as_S(mat_u8: np.ndarray, width: int) -> np.ndarray:
    """(n, w) uint8 → 'S<width>' array (NUL padded; width > w leaves room for the terminator)"""
    n, w = mat_u8.shape
    out = np.zeros((n, width), dtype=np.uint8)
    out[:, :w] = mat_u8
    return out.view("S%d" % width).reshape(n)


def make_lists(n_barcodes: int, n_genes: int, seed: int = 926, gene_prefix: str = "ENSG", gene_stride: int = 1,
               gene_start: int = 1):
    """distinct random 16-mer barcodes + '-1'; genes '<prefix>%011d' (ids i * gene_stride + 1: a stride above 1 gives
    the sparse id range of real Ensembl lists).  Returns (barcodes_text, features_text, barcodes 'S19', genes 'S<k>')."""
    rng = np.random.default_rng(seed)
    codes = np.unique(rng.integers(0, 1 << 32, size=int(n_barcodes * 1.2) + 16, dtype=np.uint64))
    rng.shuffle(codes)
    assert len(codes) >= n_barcodes
    codes = codes[:n_barcodes]
    m = _kmers(codes, 16)
    suffix = np.tile(np.frombuffer(b"-1", dtype=np.uint8), (n_barcodes, 1))
    bar = _as_S(np.concatenate([m, suffix], axis=1), 19)
    genes = np.array([("%s%011d" % (gene_prefix, i * gene_stride + gene_start)).encode() for i in range(n_genes)])
    glen = genes.dtype.itemsize + 1
    genes = genes.astype("S%d" % glen)
    barcodes_text = b"".join(b + b"\n" for b in bar.tolist())
    features_text = b"".join(g + b"\tGene" + str(i + 1).encode() + b"\tGene Expression\n"
                             for i, g in enumerate(genes.tolist()))
    return barcodes_text, features_text, bar, genes


def make_records(n: int, bar: np.ndarray, genes: np.ndarray, seed: int = 1,
                 umi_len: int = 10, umi_pool: int = 0, cell_dist: str = "uniform", gene_dist: str = "uniform",
                 p_no_cb: float = 0.0, p_unlisted_cb: float = 0.0, p_bad_xf: float = 0.0,
                 p_n_umi: float = 0.0, p_multi_gene: float = 0.0, p_no_ub: float = 0.0,
                 dup_factor: float = 0.0, zipf_umi: float = 0.0):
    """String-level SoA records: (flags u8, xf i32, cb S, gx S, ub S).

    Every record that carries a CB carries xf, and every xf in {25,17} record carries GX
    (the reference dereferences them unchecked, bam2db_ds.c:394-395,403-404)."""
    rng = np.random.default_rng(seed)
    nb, ng = len(bar), len(genes)

    def pick(dist, size, k):
        if dist == "uniform":
            return rng.integers(0, k, size=size)
        if dist == "lognormal":
            w = rng.lognormal(0.0, 1.0, size=k)
            return rng.choice(k, size=size, p=w / w.sum())
        if dist == "zipf":
            w = 1.0 / np.arange(1, k + 1) ** 1.1
            return rng.choice(k, size=size, p=w / w.sum())
        raise ValueError(dist)

    if dup_factor > 1.0:
        # 10x-like duplication: draw n/dup distinct molecules, then resample them
        n_mol = max(1, int(n / dup_factor))
        mol_cell = pick(cell_dist, n_mol, nb)
        mol_gene = pick(gene_dist, n_mol, ng)
        mol_umi = rng.integers(0, 1 << (2 * umi_len), size=n_mol, dtype=np.uint64)
        src = rng.integers(0, n_mol, size=n)
        cell, gene, ucode = mol_cell[src], mol_gene[src], mol_umi[src]
    else:
        cell = pick(cell_dist, n, nb)
        gene = pick(gene_dist, n, ng)
        if zipf_umi > 0 and umi_pool > 0:
            w = 1.0 / np.arange(1, umi_pool + 1) ** zipf_umi
            pool = rng.integers(0, 1 << (2 * umi_len), size=(ng if ng < 4096 else 4096, umi_pool), dtype=np.uint64)
            pi = rng.choice(umi_pool, size=n, p=w / w.sum())
            ucode = pool[gene % pool.shape[0], pi]
        elif umi_pool > 0:
            pool = rng.integers(0, 1 << (2 * umi_len), size=umi_pool, dtype=np.uint64)
            ucode = pool[rng.integers(0, umi_pool, size=n)]
        else:
            ucode = rng.integers(0, 1 << (2 * umi_len), size=n, dtype=np.uint64)

    cb = bar[cell].copy()
    gx = genes[gene].copy()
    ub = _as_S(_kmers(ucode, umi_len), umi_len + 1)

    flags = np.full(n, HAS_CB | HAS_XF | HAS_GX | HAS_UB, dtype=np.uint8)
    xf = np.full(n, 25, dtype=np.int32)
    xf[rng.random(n) < 0.3] = 17
    if p_bad_xf:
        bad = rng.random(n) < p_bad_xf
        xf[bad] = rng.choice(np.array([0, 1, 8, 9, 16, 24], dtype=np.int32), size=int(bad.sum()))
        nogx = bad & (rng.random(n) < 0.5)
        flags[nogx] &= ~np.uint8(HAS_GX)
    if p_unlisted_cb:
        un = rng.random(n) < p_unlisted_cb
        k = int(un.sum())
        m = _kmers(rng.integers(0, 1 << 32, size=k, dtype=np.uint64), 16)
        alt = _as_S(np.concatenate([m, np.tile(np.frombuffer(b"-9", dtype=np.uint8), (k, 1))], axis=1), bar.dtype.itemsize)
        cb[un] = alt
    if p_no_cb:
        no = rng.random(n) < p_no_cb
        flags[no] &= ~np.uint8(HAS_CB)
    if p_n_umi:
        nn = np.nonzero(rng.random(n) < p_n_umi)[0]
        ubm = ub.view(np.uint8).reshape(n, umi_len + 1)
        ubm[nn, rng.integers(0, umi_len, size=len(nn))] = ord("N")
    if p_multi_gene:
        mg = np.nonzero(rng.random(n) < p_multi_gene)[0]
        w = gx.dtype.itemsize
        wide = np.zeros(n, dtype="S%d" % (2 * w + 2))
        wide[:] = gx
        g2 = genes[rng.integers(0, ng, size=len(mg))]
        wide[mg] = np.char.add(np.char.add(gx[mg], b";"), g2)
        gx = wide
    if p_no_ub:
        nu = rng.random(n) < p_no_ub
        flags[nu] &= ~np.uint8(HAS_UB)
    return flags, xf, cb, gx, ub


def as_cstr(a: np.ndarray) -> np.ndarray:
    """make sure every element has room for a terminating NUL"""
    a = np.ascontiguousarray(a)
    longest = int(np.char.str_len(a).max()) if len(a) else 0
    if longest >= a.dtype.itemsize:
        a = a.astype("S%d" % (longest + 1))
    return a


# ---------------------------------------------------------------------------------
# BGZF / BAM writer (tests only need unmapped records with aux tags)
# ---------------------------------------------------------------------------------
def _bgzf_block(payload: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = co.compress(payload) + co.flush()
    bsize = len(data) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + data + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bam_record(name: bytes, aux: bytes, seq_len: int = 0, n_cigar: int = 0, ref_id: int = -1, pos: int = -1) -> bytes:
    """one alignment record; seq_len / n_cigar > 0 give the shape of a mapped read (CIGAR words, packed bases, qualities
    between the read name and the tags)"""
    flag = 4 if ref_id < 0 else 0
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name) + 1, 255 if ref_id >= 0 else 0, 4680, n_cigar, flag, seq_len, -1, -1, 0)
    body += name + b"\0" + struct.pack("<%dI" % n_cigar, *([(seq_len << 4) | 0] * n_cigar))
    body += bytes([0x12] * ((seq_len + 1) // 2)) + bytes([30] * seq_len) + aux
    return struct.pack("<i", len(body)) + body


def aux_Z(tag: bytes, val: bytes, typ: bytes = b"Z") -> bytes:
    """string value; typ b"H" writes the same bytes under the hex-string type (htslib's bam_aux2Z accepts both)"""
    return tag + typ + val + b"\0"


def aux_int(tag: bytes, val: int, typ: bytes = b"C") -> bytes:
    fmt = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}[typ]
    return tag + typ + struct.pack(fmt, val)


def write_bam(path: str, flags, xf, cb, gx, ub, xf_type: bytes = b"C", extra_aux=None, header_text: bytes = b"@HD\tVN:1.6\n",
              refs=(("chr1", 1000),), shape=None, str_type=None):
    """records → BGZF BAM file.  extra_aux(i) may return bytes inserted BEFORE the tags of record i;
    shape(i) may return (seq_len, n_cigar, ref_id, pos) for a mapped-read layout; str_type(i) may return the type
    byte (b"Z" / b"H") of record i's CB, GX and UB values."""
    out = bytearray()
    payload = bytearray(b"BAM\1" + struct.pack("<i", len(header_text)) + header_text + struct.pack("<i", len(refs)))
    for nm, ln in refs:
        nb = nm.encode() + b"\0"
        payload += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)

    def flush(force=False):
        nonlocal payload
        while len(payload) >= 0xff00 or (force and payload):
            out.extend(_bgzf_block(bytes(payload[:0xff00])))
            payload = payload[0xff00:]

    cbl, gxl, ubl = cb.tolist(), gx.tolist(), ub.tolist()
    for i in range(len(flags)):
        aux = bytearray()
        if extra_aux is not None:
            aux += extra_aux(i)
        f = int(flags[i])
        st = str_type(i) if str_type is not None else b"Z"
        if f & HAS_CB:
            aux += aux_Z(b"CB", cbl[i], st)
        if f & HAS_XF:
            aux += aux_int(b"xf", int(xf[i]), xf_type)
        if f & HAS_GX:
            aux += aux_Z(b"GX", gxl[i], st)
        if f & HAS_UB:
            aux += aux_Z(b"UB", ubl[i], st)
        payload += bam_record(b"r%d" % i, bytes(aux), *(shape(i) if shape is not None else ()))
        if len(payload) >= 0xff00:
            flush()
    flush(force=True)
    out.extend(_BGZF_EOF)
    with open(path, "wb") as fh:
        fh.write(out)
