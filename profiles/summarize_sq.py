"""Reduce the SQ counter passes of profiles/collect.sh to <tag>_sq_counters.json: per kernel, counters per
wave (cycle counters x 4: SQ cycle counters tick once per 4 shader clocks on gfx9) and the derived
issue-utilisation figures DESIGN.md quotes."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
CYC = {"SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
       "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CYCLES"}


def short(name):
    m = re.search(r"\bk_\w+(<[^>(]*>)?", name)
    return m.group(0) if m else name


kern = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "sq*"))):
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r.get("Dispatch_Id"))
    for k, v in acc.items():
        w = max(v.get("SQ_WAVES", 1.0), 1.0)
        for c, x in v.items():
            if c != "SQ_WAVES":
                kern[k][c + "_per_wave"] = (4.0 if c in CYC else 1.0) * x / w
        kern[k]["waves"] = w
        kern[k]["launches"] = max(1, len(disp[k]))
for k, v in kern.items():
    wc = v.get("SQ_WAVE_CYCLES_per_wave")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c + "_per_wave" in v:
                v[c.lower().replace("sq_", "frac_")] = v[c + "_per_wave"] / wc
workload = None
try:
    cfg = json.load(open(os.path.join(out, "fetch.json")))["config"]
    workload = "%s, %d ints, %s, block %d, ckpt %d" % (cfg["codec"], cfg["ints_per_gpu"], cfg["distribution"], cfg["block_ints"], cfg["ckpt_interval"])
except Exception:  # noqa: BLE001
    pass
doc = {"workload": workload, "note": "rocprofv3 --kernel-trace --pmc <SQ set> (three separate passes), python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu --no-profile --no-extra; sums over all launches of a kernel divided by its wave count; cycle "
               "counters multiplied by 4 (SQ counts in units of 4 shader clocks)",
       "kernels": kern}
json.dump(doc, open(os.path.join(out, tag + "_sq_counters.json"), "w"), indent=1)
for k in ("k_encode<1>", "k_decode_rank<false, true>"):
    if k in kern:
        print(k, {a: round(b, 3) for a, b in kern[k].items() if a.startswith("frac_") or a.startswith("SQ_INSTS")})
