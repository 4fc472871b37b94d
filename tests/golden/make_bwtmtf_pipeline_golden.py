#!/usr/bin/env python3
"""Pins tools/generate_bwtmtf (suffix array -> BWT -> move-to-front) to the reference's own pipeline.  Run in the
authoring container only:

    python tests/golden/make_bwtmtf_pipeline_golden.py

Every case is a parsed text T (integers; word ids numbered by first appearance, or byte values) and the ranks
`ref_bwtmtf` (oracle/ref_shim.cpp: the statements of /root/reference/src/generate_bwtmtf.cpp:142-173 around the
UNMODIFIED include/qsufsort.hpp) produces for it.  Only integers are committed (tests/golden/bwtmtf_pipeline.json);
tests/test_bwtmtf.py rebuilds a text file from T, runs the package's tool on it and compares both of its outputs."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def first_appearance_ids(seq):
    ids, out = {}, []
    for x in seq:
        if x not in ids:
            ids[x] = len(ids) + 1
        out.append(ids[x])
    return out


def cases():
    rng = np.random.default_rng(20260104)
    out = []
    z = np.minimum(rng.zipf(1.3, size=6000), 800).tolist()
    out.append(("word: Zipf(1.3) over 800 words, 6000 words", "word", first_appearance_ids(z), 6000))
    per = [3, 1, 4, 1, 5, 9, 2] * 500
    for i in rng.integers(0, len(per), size=40):
        per[int(i)] = int(rng.integers(10, 60))
    out.append(("word: period-7 pattern with 40 substitutions, 3500 words", "word", first_appearance_ids(per), 3500))
    out.append(("word: one word 300 times", "word", [1] * 300, 300))
    out.append(("word: truncated by -n (5000 words, first 2000 kept)", "word", first_appearance_ids(np.minimum(rng.zipf(1.5, size=5000), 300).tolist())[:2000], 2000))
    out.append(("byte: uniform bytes 1..255, 5000", "byte", rng.integers(1, 256, size=5000).tolist(), 5000))
    out.append(("byte: 'abracadabra ' repeated with noise, 4000", "byte",
                [c if rng.random() > 0.01 else int(rng.integers(97, 123)) for c in (b"abracadabra " * 400)[:4000]], 4000))
    out.append(("byte: two-letter alphabet, long runs, 3000", "byte", (np.repeat(rng.integers(97, 99, size=300), rng.integers(1, 30, size=300))[:3000]).tolist(), 3000))
    return out


def main():
    assert ol.have_ref() and hasattr(ol.ref(), "ref_bwtmtf"), "build oracle/_ref first (make -C oracle)"
    doc = {"source": "ref_bwtmtf in oracle/ref_shim.cpp: generate_bwtmtf.cpp:142-173 around the unmodified qsufsort.hpp", "cases": []}
    for name, mode, T, n in cases():
        T = [int(x) for x in T]
        mtf = ol.ref_bwtmtf(T + [0], n)
        assert mtf.size == min(len(T), n)
        doc["cases"].append({"name": name, "mode": mode, "n": n, "T": T, "mtf": [int(x) for x in mtf]})
        print("%-60s %5d ranks, max %d" % (name, mtf.size, int(mtf.max())))
    with open(os.path.join(HERE, "bwtmtf_pipeline.json"), "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))


if __name__ == "__main__":
    main()
