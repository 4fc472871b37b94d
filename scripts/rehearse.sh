#!/bin/bash
# N > 1 control-flow rehearsal of bench.py on ONE GPU (gloo + host copies; not a measurement)
O=gpurun_out/${1:-reh}; mkdir -p $O
for cfg in "2 fixed" "3 fixed" "3 rotate"; do
  set -- $cfg; w=$1; root=$2
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2951$w bench.py --gpus $w --steps 4 --warmup 2 --ints 4194304 --rehearse-gloo --no-cpu --gather-root $root > $O/rehearse_${w}_$root.json 2> $O/rehearse_${w}_$root.err || { echo "rehearsal $cfg FAILED"; tail -15 $O/rehearse_${w}_$root.err; }
  python - $O/rehearse_${w}_$root.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("rehearse", d["n_gpus"], "ranks ok=%s merged_ok=%s %.1f Mints/s :: %s"%(d["roundtrip_ok"],d["merged_container_ok"],d["value"],d["config"]["multi_gpu"][:90]))
except Exception as e: print("rehearsal output unreadable", e)
PY
done
