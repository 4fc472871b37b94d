"""Generates tests/golden/zipf_trace.json from the reference's zipf_distribution (include/zipf_dist.hpp) compiled into
oracle/_ref/libans_ref.so: per (n, q), values drawn with std::mt19937(seed) and the canonical uniforms each one
consumed (rejected draws first, the accepted one last), as hex floats.  Run in the authoring container:
    python tests/golden/make_zipf_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

doc = {"note": "reference zipf_distribution<uint32_t>(n, q) on std::mt19937(seed): values, draws per value, canonical "
               "uniforms (generate_canonical<double,53>) in consumption order", "cases": []}
for n, q, seed in ((1 << 12, 1.0, 0), (1 << 20, 1.0, 0), (1 << 20, 1.2, 1), (1 << 24, 1.2, 2), (1 << 24, 1.0, 3), ((1 << 30) - 1, 2.0, 4),
                   (100, 0.5, 5)):
    vals, nd, u = ol.ref_zipf_trace(n, q, seed, 1500)
    doc["cases"].append({"n": n, "q": q, "seed": seed, "values": [int(x) for x in vals], "draws": [int(x) for x in nd],
                         "u01": [float(x).hex() for x in u]})
with open(os.path.join(HERE, "zipf_trace.json"), "w") as fh:
    json.dump(doc, fh, separators=(",", ":"))
print("wrote", sum(len(c["values"]) for c in doc["cases"]), "values")
