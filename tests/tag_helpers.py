"""Synthetic BAMs for the crb / extract tests: records that carry CB / CR / UB / GX / xf / custom tags."""
import numpy as np

from fastf_amd import synth


def random_dna(rng, n_pool, length, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [bytes(a[rng.integers(0, len(a), length)]) for _ in range(n_pool)]


class TagCase:
    """n records: CB (corrected barcode + '-1') on most, CR (raw barcode: the CB with an occasional substitution or N),
    UB, GX, xf; p_no_cb of the records carry no CB (and the reference then ignores CR)."""

    def __init__(self, n, n_cb=200, seed=1, p_no_cb=0.05, p_err=0.08, p_n=0.02, n_gene=50, umi_pool=300, cb_len=16,
                 sorted_cb=False):
        rng = np.random.default_rng(seed)
        pool = sorted(set(random_dna(rng, n_cb, cb_len)))
        pick = rng.integers(0, len(pool), n)
        if sorted_cb:
            pick = np.sort(pick)
        self.n = n
        self.has_cb = rng.random(n) >= p_no_cb
        self.cb, self.cr = [], []
        err, nn = rng.random(n) < p_err, rng.random(n) < p_n
        pos, sub = rng.integers(0, cb_len, n), rng.integers(0, 4, n)
        for i in range(n):
            b = pool[pick[i]]
            r = bytearray(b)
            if err[i]:
                r[pos[i]] = b"ACGT"[sub[i]]
            if nn[i]:
                r[(pos[i] * 7) % cb_len] = ord("N")
            self.cb.append(b + b"-1")
            self.cr.append(bytes(r))
        umis = random_dna(rng, umi_pool, 10)
        self.ub = [umis[j] for j in rng.integers(0, umi_pool, n)]
        self.gx = [b"ENSG%011d" % (j + 1) for j in rng.integers(0, n_gene, n)]
        self.gn = [b"GENE-%d.%d" % (j % 17, j) for j in rng.integers(0, n_gene, n)]
        self.xf = rng.choice(np.array([25, 17, 0, 1, 19]), n)
        self.nh = rng.integers(-3, 400, n)                 # an integer tag with negative and multi-digit values
        self.has_gx = rng.random(n) < 0.8

    def write(self, path):
        z = np.zeros(self.n, dtype=np.uint8)
        e = np.array([b""] * self.n, dtype="S1")

        def aux(i):
            a = bytearray()
            a += synth.aux_Z(b"CR", self.cr[i])
            if self.has_cb[i]:
                a += synth.aux_Z(b"CB", self.cb[i])
            a += synth.aux_Z(b"UB", self.ub[i])
            if self.has_gx[i]:
                a += synth.aux_Z(b"GX", self.gx[i]) + synth.aux_Z(b"GN", self.gn[i])
            a += synth.aux_int(b"xf", int(self.xf[i]), b"C")
            a += synth.aux_int(b"NH", int(self.nh[i]), b"i" if self.nh[i] < 0 or self.nh[i] > 127 else b"c")
            return bytes(a)
        synth.write_bam(path, z, z.astype(np.int32), e, e, e, extra_aux=aux)

    @staticmethod
    def S(strings):
        w = max([len(s) for s in strings] + [1]) + 1
        return np.array(strings, dtype="S%d" % w)
