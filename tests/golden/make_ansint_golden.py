"""Generates tests/golden/ansint.json: plain ANSint (methods.hpp:484-497 -> ans_int_compress) streams made by the real
reference (oracle/_ref) on lists whose values fit the GPU path's 16384-symbol model, including low-entropy lists whose
frame exceeds 2^16 (32-bit frequencies, ans_int.hpp:30-34).  Run in the authoring container:
    python tests/golden/make_ansint_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def sparse_list(n, pairs):
    d = np.zeros(n, dtype=np.uint32)
    for pos, val in pairs:
        d[pos] = val
    return d


def main():
    rows = []
    for fam in ("uniform256", "geom0.01", "geom0.4", "uniform12", "zipf20s1.2", "boundaries"):
        for n in (7, 1000, 65536, 300001):
            seed = 300 + n % 97
            d = np.minimum(ol.gen_inputs(fam, n, seed), np.uint32(16383))
            raw = ol.ref_encode(ol.INT, 0, d)
            s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
            assert raw.size == s.size and np.array_equal(ol.ref_decode(ol.INT, 0, raw, n), d)
            assert np.array_equal(ol.canonicalize(raw, info), ol.canonicalize(s, info))
            rows.append({"input": "family", "family": fam, "n": n, "seed": seed, "clip": 16383, "log2_frame": int(info.log2_frame),
                         "stream_len": int(raw.size), "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                         "stream_sha256": hashlib.sha256(ol.canonicalize(raw, info).tobytes()).hexdigest()})
    # almost-constant lists: tiny entropy -> the frame-size search runs far beyond 2^16
    rng = np.random.default_rng(20261004)
    for n, k, vmax in ((200000, 40, 2000), (600000, 7, 16000), (70000, 3, 16383), (1 << 20, 120, 300)):
        pairs = [(int(p), int(v)) for p, v in zip(rng.choice(n, size=k, replace=False), rng.integers(1, vmax + 1, size=k))]
        d = sparse_list(n, pairs)
        raw = ol.ref_encode(ol.INT, 0, d)
        s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
        assert np.array_equal(ol.ref_decode(ol.INT, 0, raw, n), d)
        assert np.array_equal(ol.canonicalize(raw, info), ol.canonicalize(s, info))
        rows.append({"input": "sparse", "n": n, "pairs": pairs, "log2_frame": int(info.log2_frame), "stream_len": int(raw.size),
                     "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                     "stream_hex": ol.canonicalize(raw, info).tobytes().hex()})
    with open(os.path.join(HERE, "ansint.json"), "w") as fh:
        json.dump(rows, fh, indent=0)
    assert sum(1 for r in rows if r["log2_frame"] > 16) >= 2
    print("ansint.json:", len(rows), "entries; frames", sorted(set(r["log2_frame"] for r in rows)))
    # ansint_large.json (round 4): values far beyond 16384 -- the reference sizes its model by the largest value
    # (ans_int.hpp:41-51) and its prelude codes max_sym + 1 items; the GPU path models such a list in rank space
    # (csrc/ansx_intsparse.h) and must write the same bytes.  Lists of at most 16384 ints (one block's worth).
    big = []
    for shape in ("uniform", "skew", "cluster"):
        for n in (7, 1000, 5000, 16384):
            for lg in (17, 20, 22):
                seed = 4000 + n % 89 + lg
                d = ol.ansint_large_list(n, 1 << lg, seed, shape)
                raw = ol.ref_encode(ol.INT, 0, d)
                s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
                assert raw.size == s.size and np.array_equal(ol.ref_decode(ol.INT, 0, raw, n), d)
                assert np.array_equal(ol.canonicalize(raw, info), ol.canonicalize(s, info))
                row = {"shape": shape, "n": n, "log2_vmax": lg, "seed": seed, "log2_frame": int(info.log2_frame),
                       "prelude_bytes": int(info.prelude_bytes), "stream_len": int(raw.size),
                       "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                       "stream_sha256": hashlib.sha256(ol.canonicalize(raw, info).tobytes()).hexdigest()}
                if n <= 1000:
                    row["stream_hex"] = ol.canonicalize(raw, info).tobytes().hex()
                big.append(row)
    # lists longer than one block's worth: what bounds the rank-space model is the number of DISTINCT values (16384)
    for shape, n, lg in (("cluster", 100000, 20), ("cluster", 300001, 22), ("cluster", 65536, 17)):
        seed = 4100 + lg
        d = ol.ansint_large_list(n, 1 << lg, seed, shape)
        raw = ol.ref_encode(ol.INT, 0, d)
        s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
        assert raw.size == s.size and np.array_equal(ol.ref_decode(ol.INT, 0, raw, n), d)
        assert np.array_equal(ol.canonicalize(raw, info), ol.canonicalize(s, info))
        big.append({"shape": shape, "n": n, "log2_vmax": lg, "seed": seed, "log2_frame": int(info.log2_frame),
                    "prelude_bytes": int(info.prelude_bytes), "stream_len": int(raw.size), "distinct": int(np.unique(d).size),
                    "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                    "stream_sha256": hashlib.sha256(ol.canonicalize(raw, info).tobytes()).hexdigest()})
    with open(os.path.join(HERE, "ansint_large.json"), "w") as fh:
        json.dump(big, fh, indent=0)
    print("ansint_large.json:", len(big), "entries; preludes", min(r["prelude_bytes"] for r in big), "..", max(r["prelude_bytes"] for r in big), "bytes")


if __name__ == "__main__":
    main()
