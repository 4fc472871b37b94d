"""Golden vectors for fidelities 6 and 7 (alphabets of 32 Ki / 64 Ki slots), generated from the real reference
(oracle/_ref) like f24.json: hashed inputs and streams.

    make -C oracle && python tests/golden/make_f67_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_lib as ol  # noqa: E402
from make_golden import FAMILIES, entry  # noqa: E402


def main():
    assert ol.have_ref(), "build oracle/_ref first (make -C oracle)"
    out = []
    for k in ("fold", "rfold"):
        for f in (6, 7):
            for fam in FAMILIES:
                for n in (5, 313, 4096, 70001):
                    if fam == "distinct" and n > 65536:
                        continue
                    seed = 1000 * f + n
                    d = ol.gen_inputs(fam, n, seed)
                    if k == "rfold":
                        d = d % np.uint32(1 << 21)  # reference rfold allocates 16*(max+1) bytes
                    out.append(entry(k, f, d, False, fam, seed))
    with open(os.path.join(HERE, "f67.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("f67:", len(out))


if __name__ == "__main__":
    main()
