"""Reduce the rocprofv3 outputs of profiles/collect.sh to the two committed summaries:
<tag>_kernel_stats.csv (ansx kernels only) and <tag>_hbm_traffic_pmc.json (bytes per launch).
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM
section; calibrated on k_fold_hist, which reads exactly 4n bytes)."""
import collections
import csv
import glob
import json
import re
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"\bk_\w+(<[^>(]*>)?", name)
    return m.group(0) if m else name


stats = glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv"))[0]
rows = [r for r in csv.DictReader(open(stats)) if short(r["Name"]).startswith("k_")]
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        r["Name"] = short(r["Name"])
        w.writerow(r)


def counter(which, cname):
    cc = glob.glob(os.path.join(out, which, "*", "*_counter_collection.csv"))[0]
    acc, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_") or r["Counter_Name"] != cname:
            continue
        acc[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(launches[k]) for k in acc}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
cfg = json.load(open(os.path.join(out, "fetch.json")))["config"]
kern = {}
for k in fetch:
    fb, wb = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    kern[k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0), "fetch_bytes": fb,
               "write_bytes": wb, "hbm_bytes": fb + wb}
doc = {
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py "
            "--steps 1 --warmup 0 --no-cpu --no-profile; per-launch averages; units KB; fetch_bytes = "
            "FETCH_SIZE*1024*2 (gfx950 correction of MI355X_MICROARCH.md; check: k_fold_hist reads exactly 4n "
            "bytes), write_bytes = WRITE_SIZE*1024",
    "workload": "%s, %d ints, %s, block %d, ckpt %d" % (cfg["codec"], cfg["ints_per_gpu"], cfg["distribution"],
                                                      cfg["block_ints"], cfg["ckpt_interval"]),
    "kernels": kern,
}
json.dump(doc, open(os.path.join(out, tag + "_hbm_traffic_pmc.json"), "w"), indent=1)
for k, v in kern.items():
    print("%-28s fetch %8.1f MB  write %8.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))
print(open(os.path.join(out, "bench.json")).read()[:600])
