"""Generates tests/golden/config0_digest.json: digests of the oracle's outputs on BASELINE configs[0] (100 k synthetic
records, 1 k barcodes x 500 genes, --cell 1.0 --depth 1.0 --seed 926) and on its --cell 0.5 --depth 0.5 variant, from the
seeded generator in fastf_amd/synth.py.  A regression anchor for oracle + generator (the GPU path is compared with the
oracle itself in tests/test_gpu_parity.py):  python tests/golden/make_config0_digest.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import Case  # noqa: E402

out = {}
for name, kw in (("config0_keepall", dict(n=100_000, n_bar=1000, n_gene=500, umi_pool=64)),
                 ("config0_half", dict(n=100_000, n_bar=1000, n_gene=500, umi_pool=64, rate_cell=0.5, rate_depth=0.5))):
    o = Case(**kw).oracle()
    out[name] = {"kw": kw, "counters": [o["total"], o["sampled"], o["valid"]], "nnz": o["nnz"],
                 "matrix_md5": hashlib.md5(o["matrix"]).hexdigest(), "barcodes_md5": hashlib.md5(o["barcodes"]).hexdigest(),
                 "features_md5": hashlib.md5(o["features"]).hexdigest(), "umi_md5": hashlib.md5(o["umi"]).hexdigest(),
                 "matrix_head": o["matrix"][:200].decode()}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config0_digest.json"), "w"), indent=1)
print("wrote config0_digest.json")
