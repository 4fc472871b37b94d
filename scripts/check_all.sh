#!/bin/bash
# full GPU check: parity suite, default bench line (with extra_configs), N > 1 rehearsal on one GPU
set -o pipefail
O=gpurun_out/${1:-chk}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d.get("details"):
    try: d.update(json.load(open(d["details"])))
    except OSError: pass
def row(tag,r):
    k=r.get("kernels") or {}
    print("%-28s %.1f Gints/s %.3f ms bpi %.3f ok=%s near=%s path=%s roof=%s %.3f | %s"%(tag,r["value"]/1e3,r["ms_per_step"],r["bits_per_int"],r["roundtrip_ok"],r.get("near_threshold_decisions"),r.get("encode_path"),r["roofline"]["kernel"],r["roofline"]["frac"]," ".join("%s=%.3f"%(n.replace("k_",""),v["avg_ms"]*v["launches_per_step"]) for n,v in k.items())))
    c=r.get("cpu_baseline_full") or r.get("cpu_baseline")
    if c and "blocked" in c: print("   cpu %s %.1f Mints/s (enc %.1f dec %.1f) bpi %.3f blocked %.1f Mints/s bpi %.3f"%(c["kind"],c["value"],c["enc_mints"],c["dec_mints"],c["bits_per_int"],c["blocked"]["value"],c["blocked"]["bits_per_int"]))
row("main",d)
for e in d.get("extra_configs") or []:
    if "error" in e: print(e["baseline_config"],"ERROR",e["error"])
    else: row(e["baseline_config"][:28],e)
PY
bash scripts/rehearse.sh ${1:-chk}
