"""Generates tests/golden/tag_tree.json: outputs of the REFERENCE's own tree code (filter.c insert_tree,
print_tree, print_tree_same_row, compiled in place into oracle/_ref/libfastf_ref_tree.so) on seeded tag lists.
Run in the build container (needs /root/reference):  python tests/golden/make_tag_tree.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from test_oracle_tag_pins import _ref_print_tree, _ref_same_row, _random_tags  # noqa: E402

R = O.ref_tree_lib()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
out = {"extract": [], "same_row": []}
for seed, n, pool, alpha in [(11, 60, 12, b"ACGT"), (12, 200, 40, b"ACGTN"), (13, 40, 40, b"0123456789-")]:
    tags = _random_tags(np.random.default_rng(seed), n, pool, alphabet=alpha, length=6)
    out["extract"].append({"tags": [t.decode() for t in tags], "print_tree": _ref_print_tree(R, tags).decode()})
for seed, n, pool in [(21, 80, 9), (22, 30, 30)]:
    tags = _random_tags(np.random.default_rng(seed), n, pool, alphabet=b"ACGTN", length=5)
    out["same_row"].append({"tags": [t.decode() for t in tags], "print_tree_same_row": _ref_same_row(R, tags).decode()})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "tag_tree.json"), "w"), indent=0)
print("wrote tag_tree.json")
