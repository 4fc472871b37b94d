#!/bin/bash
# quick bench summary: scripts/quick.sh <tag> [bench args]
O=gpurun_out/q; mkdir -p $O; tag=$1; shift
python bench.py --steps 10 --warmup 2 --no-cpu "$@" > $O/$tag.json 2> $O/$tag.err || { tail -2 $O/$tag.err; [ -s $O/$tag.json ] || exit 1; }
grep -v "amdgpu.ids" $O/$tag.err | tail -3
python - $O/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d.get("details"):
    try: d.update(json.load(open(d["details"])))
    except OSError: pass
k=d.get("kernels") or {}
print("%.1f Gints/s %.3f ms bpi %.3f ok=%s |"%(d["value"]/1e3,d["ms_per_step"],d["bits_per_int"],d["roundtrip_ok"]), " ".join("%s=%.3f"%(n.replace("k_",""),v["avg_ms"]*v["launches_per_step"]) for n,v in k.items()))
PY
