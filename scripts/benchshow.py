#!/usr/bin/env python3
"""Print the per-kernel picture of a bench.py JSON line (development helper)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d.get("details"):  # per-kernel tables and the extra configurations live in the side file since round 4
    try:
        d.update(json.load(open(d["details"])))
    except OSError:
        pass


def show(e, label):
    print("%s: %.1f Mints/s  %.3f ms/step  ok=%s path=%s  %.3f bits/int" % (label, e["value"], e["ms_per_step"], e["roundtrip_ok"],
                                                                          e.get("encode_path"), e["bits_per_int"]))
    r = e.get("roofline") or {}
    print("   roofline: %s %.3f (traffic %s)" % (r.get("kernel"), r.get("frac", 0), r.get("traffic")))
    if e.get("kernels"):
        tot = 0.0
        for k, v in e["kernels"].items():
            print("   %-20s %.3f ms x%.1f" % (k, v["avg_ms"], v["launches_per_step"]))
            tot += v["avg_ms"] * v["launches_per_step"]
        print("   sum of kernels %.3f ms" % tot)


show(d, "main")
for e in d.get("extra_configs") or []:
    if "error" in e:
        print(e)
        continue
    show(e, e["baseline_config"])
