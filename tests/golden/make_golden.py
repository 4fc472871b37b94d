#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libans_ref.so, built by
oracle/Makefile from the unmodified headers under /root/reference/include).

Run in the authoring container only:  python tests/golden/make_golden.py
Fixtures are data: inputs (verbatim for small n, generator name + seed + sha256 for large n) and
the expected reference byte stream with the indeterminate padding bits of the last
interpolative word zeroed (`interp_bits` says how many bits of the code are valid).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

SMALL_N = [1, 2, 3, 5, 7, 64, 313, 1000, 1001]
LARGE_N = [4096, 65536, 300007]
FAMILIES = ["uniform256", "uniform12", "geom0.01", "geom0.4", "zipf20s1.2", "constant",
            "distinct", "sparse_large", "boundaries"]
EXPLICIT = [  # SURVEY 8c canonical known answers + F3 triggers
    ("fold", 1, [1, 2, 3]),
    ("fold", 1, [5, 300]),
    ("fold", 1, [1, 2, 3, 70000, 5]),
    ("rfold", 1, [1, 2, 3, 70000, 5]),
    ("fold", 3, [9] * 7 + [2]),
    ("fold", 1, [5]),
    ("fold", 1, [7] * 1000),
    ("rfold", 1, [5, 1000, 5, 70000, 5, 5, 1000]),
    ("fold", 1, [(1 << 30) - 1, 0, 255, 256, 65535, 65536, 16777215, 16777216]),
    ("msb", 0, [(1 << 30) - 1, 0, 255, 256, 257, 65535, 65536, 65537, 16777215, 16777216, 16777217]),
    ("msb", 0, [5]),
    ("msb", 0, [300, 300, 7]),
]
KIND = {"fold": ol.FOLD, "rfold": ol.RFOLD, "msb": ol.MSB}


def entry(kind_name, f, data, verbatim, name=None, seed=None):
    kind = KIND[kind_name]
    data = np.ascontiguousarray(data, dtype=np.uint32)
    raw = ol.ref_encode(kind, f, data)
    # valid-bit count: from the restatement, cross-checked against the reference's own byte count
    s, info, _, _ = ol.oracle_encode(kind, f, data)
    assert len(s) == len(raw)
    canon = ol.canonicalize(raw, info)
    dec = ol.ref_decode(kind, f, raw, data.size)
    e = {
        "kind": kind_name, "f": f, "n": int(data.size),
        "header_bytes": info.header_bytes, "prelude_bytes": info.prelude_bytes,
        "interp_bits": info.interp_bits, "max_sym": info.max_sym, "log2_frame": info.log2_frame,
        "reorder_flag": info.reorder_flag, "stream_len": int(len(raw)),
        "ref_roundtrip_ok": bool(np.array_equal(dec, data)),
    }
    if name is not None:
        e["family"] = name
        e["seed"] = seed
    if verbatim:
        e["input"] = [int(x) for x in data]
        e["stream_hex"] = canon.tobytes().hex()
    else:
        e["input_sha256"] = hashlib.sha256(data.tobytes()).hexdigest()
        e["stream_sha256"] = hashlib.sha256(canon.tobytes()).hexdigest()
    return e


def main():
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    small, large = [], []
    for k, f, d in EXPLICIT:
        small.append(entry(k, f, d, True))
    for k, flist in (("fold", (1, 3, 5)), ("rfold", (1, 3, 5)), ("msb", (0,))):
        for f in flist:
            for fam in FAMILIES:
                for n in SMALL_N:
                    if n > 313 and fam not in ("uniform256", "zipf20s1.2", "geom0.01", "sparse_large"):
                        continue
                    seed = 1000 * f + n
                    d = ol.gen_inputs(fam, n, seed)
                    if k == "rfold":
                        d = d % np.uint32(1 << 21)  # reference rfold allocates 16*(max+1) bytes
                    small.append(entry(k, f, d, True, fam, seed))
                for n in LARGE_N:
                    if fam in ("distinct",) and n > 65536:
                        continue
                    seed = 1000 * f + n
                    d = ol.gen_inputs(fam, n, seed)
                    if k == "rfold":
                        d = d % np.uint32(1 << 21)
                    large.append(entry(k, f, d, False, fam, seed))
    # fidelities between the reference harness's own choices (every accepted value is pinned)
    f24 = []
    for k in ("fold", "rfold"):
        for f in (2, 4):
            for fam in FAMILIES:
                for n in (5, 313, 4096, 70001):
                    if fam == "distinct" and n > 65536:
                        continue
                    seed = 1000 * f + n
                    d = ol.gen_inputs(fam, n, seed)
                    if k == "rfold":
                        d = d % np.uint32(1 << 21)
                    f24.append(entry(k, f, d, False, fam, seed))
    with open(os.path.join(HERE, "f24.json"), "w") as fh:
        json.dump(f24, fh, indent=0)
    with open(os.path.join(HERE, "small.json"), "w") as fh:
        json.dump(small, fh, separators=(",", ":"))
    with open(os.path.join(HERE, "large.json"), "w") as fh:
        json.dump(large, fh, indent=0)
    print("small:", len(small), "large:", len(large), "f24:", len(f24))


if __name__ == "__main__":
    main()
