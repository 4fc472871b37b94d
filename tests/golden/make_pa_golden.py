#!/usr/bin/env python3
"""Fixtures for per-block alphabet compaction and ANSint, generated from the REAL reference
(oracle/_ref: ref_pa_encode restates src/pseudo_adaptive.cpp:85-130 around the unmodified
interpolative_internal::encode / ANSint / ANSmsb / ANSfold functions; ref_encode kind 3 = ans_int_compress).

Run in the authoring container only:  python tests/golden/make_pa_golden.py
Writes tests/golden/pa.json: per (codec, family, n) the canonical (padding-zeroed) stream hash."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

KIND = {"fold": ol.FOLD, "msb": ol.MSB, "int": ol.INT}
FAMS = ["zipf20s1.2", "uniform256", "geom0.01", "geom0.4", "uniform12", "constant"]


def main():
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    rows = []
    for kname, f in (("msb", 0), ("int", 0), ("fold", 1), ("fold", 3)):
        for fam in FAMS:
            for n in (5, 313, 4096, 8192, 16380):
                seed = 77 + n
                d = ol.gen_inputs(fam, n, seed)
                raw, hb = ol.ref_pa_encode(KIND[kname], f, d)
                s, pinfo, info, _, _ = ol.oracle_pa_encode(KIND[kname], f, d)
                assert hb == pinfo.header_bytes and raw.size == s.size
                canon = ol.canonicalize_pa(raw, pinfo, info)
                rows.append({"mode": "pa", "kind": kname, "f": f, "family": fam, "n": n, "seed": seed,
                             "sigma": int(pinfo.sigma), "header_bytes": int(hb), "stream_len": int(raw.size),
                             "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                             "stream_sha256": hashlib.sha256(canon.tobytes()).hexdigest()})
    # plain ANSint (methods.hpp:484-497) on small-valued lists
    for fam in ("uniform256", "geom0.01", "geom0.4", "uniform12"):
        for n in (7, 1000, 65536):
            seed = 99 + n
            d = ol.gen_inputs(fam, n, seed)
            if int(d.max()) > n + 1024:
                continue  # outside the restatement's range (oracle/ans_oracle.h)
            raw = ol.ref_encode(ol.INT, 0, d)
            s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
            assert raw.size == s.size and np.array_equal(ol.ref_decode(ol.INT, 0, raw, n), d)
            rows.append({"mode": "plain", "kind": "int", "f": 0, "family": fam, "n": n, "seed": seed,
                         "stream_len": int(raw.size), "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                         "stream_sha256": hashlib.sha256(ol.canonicalize(raw, info).tobytes()).hexdigest()})
    with open(os.path.join(HERE, "pa.json"), "w") as fh:
        json.dump(rows, fh, indent=0)
    print("pa.json:", len(rows), "entries")


if __name__ == "__main__":
    main()
