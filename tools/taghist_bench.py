"""diagnostic: device tag histogram throughput (crb / extract seam) on 10 M records: single tag with few / many distinct
values, and the CB -> CR pair histogram; host keys are already packed (push = staging + H2D, finish = the device work)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from fastf_amd.tags import TagHist
N = 10_000_000
rng = np.random.default_rng(1)
def dna_keys(codes, nb=16):        # DNA-form keys: form 1, length nb, no suffix
    return (np.uint64(1) << np.uint64(62)) | (np.uint64(nb) << np.uint64(57)) | (codes.astype(np.uint64) << np.uint64(48 - 2 * nb))
cb_pool = rng.integers(0, 1 << 32, 10_000, dtype=np.uint64)
cb = dna_keys(cb_pool[rng.integers(0, len(cb_pool), N)])
cr = cb.copy(); flip = rng.random(N) < 0.08
cr[flip] ^= (np.uint64(1) << (np.uint64(16) + (rng.integers(0, 32, int(flip.sum())).astype(np.uint64))))
ub = dna_keys(rng.integers(0, 1 << 20, N, dtype=np.uint64), 10)
cb[rng.random(N) < 0.05] = 0
for name, k1, k2 in (("CB (10 k distinct)", cb, None), ("UB (1 M distinct)", ub, None), ("CB -> CR pairs", cb, cr)):
    h = TagHist()
    h.push(k1[:1000], None if k2 is None else k2[:1000]); h.finish(); h.close()          # warm-up (module load)
    h = TagHist()
    t0 = time.perf_counter(); h.push(k1, k2); t1 = time.perf_counter(); r = h.finish(); t2 = time.perf_counter()
    print("%-20s push %.1f ms  finish %.1f ms  -> %d distinct%s  (%.0f M records/s device side)" % (
        name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(r["key1"]), (", %d pairs" % len(r["pair_key2"])) if "pair_key2" in r else "", N / (t2 - t1) / 1e6))
    h.close()
