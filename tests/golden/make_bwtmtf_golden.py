#!/usr/bin/env python3
"""Config-5-shaped fixture (SURVEY 8d fallback: BWT-MTF ranks synthesised from a local text, following
/root/reference/src/generate_bwtmtf.cpp:142-173).  Run in the authoring container only:

    python tests/golden/make_bwtmtf_golden.py

Input text: the first 65536 words of this image's /usr/lib/python3.10/pydoc_data/topics.py (English
reference prose; nothing of it is committed -- only the resulting integer ranks are data).  Produces
tests/golden/bwtmtf.u32 (the ranks, via tools/generate_bwtmtf.x) and bwtmtf.json (expected reference
streams from oracle/_ref: whole list + every 16 Ki block, ANSfold-1 / ANSfold-5 / ANSrfold-1)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
TOOLS = os.path.join(ROOT, "ans_large_alphabet_amd", "tools")
TEXT = "/usr/lib/python3.10/pydoc_data/topics.py"
N = 65536


def main():
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    subprocess.check_call(["make", "-s", "-C", TOOLS, "generate_bwtmtf.x"])
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call([os.path.join(TOOLS, "generate_bwtmtf.x"), "-i", TEXT, "-n", str(N), "-w",
                               "-o", os.path.join(td, "x")])
        data = np.fromfile(os.path.join(td, "x-WORD-BWTMTF.u32"), dtype=np.uint32)
    data.tofile(os.path.join(HERE, "bwtmtf.u32"))
    streams = []
    for kind, f in (("fold", 1), ("fold", 5), ("rfold", 1)):
        k = ol.FOLD if kind == "fold" else ol.RFOLD
        spans = [(0, data.size)] + [(a, min(16384, data.size - a)) for a in range(0, data.size, 16384)]
        for first, n in spans:
            part = np.ascontiguousarray(data[first:first + n])
            raw = ol.ref_encode(k, f, part)
            s, info, _, _ = ol.oracle_encode(k, f, part)
            assert len(s) == len(raw)
            canon = ol.canonicalize(raw, info)
            assert np.array_equal(ol.ref_decode(k, f, raw, n), part)
            streams.append({"kind": kind, "f": f, "first": first, "n": n, "stream_len": int(len(raw)),
                            "stream_sha256": hashlib.sha256(canon.tobytes()).hexdigest(),
                            "bits_per_int": 8.0 * len(raw) / n})
    meta = {"source": "first %d words of %s, word-parsed BWT-MTF ranks" % (N, TEXT), "n": int(data.size),
            "max": int(data.max()), "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "streams": streams}
    with open(os.path.join(HERE, "bwtmtf.json"), "w") as fh:
        json.dump(meta, fh, indent=0)
    print("ranks:", data.size, "max", data.max(), "whole-list bits/int:",
          [round(e["bits_per_int"], 3) for e in streams if e["n"] == data.size])


if __name__ == "__main__":
    main()
