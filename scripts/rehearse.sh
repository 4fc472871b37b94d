#!/bin/bash
# N > 1 control-flow rehearsal of bench.py on ONE GPU (gloo + host copies; not a measurement):
# every root policy, and auto with a forced verification failure (must fall back to the fixed root)
O=gpurun_out/${1:-reh}; mkdir -p $O
i=0
for cfg in "2 fixed" "3 fixed" "3 rotate" "3 auto" "2 auto" "3 auto --test-fail-rotate"; do
  set -- $cfg; w=$1; root=$2; extra=$3; i=$((i+1)); tag=${w}_${root}${extra:+_fail}
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29510+i)) bench.py --gpus $w --steps 4 --warmup 2 --ints 4194304 --rehearse-gloo --no-cpu --gather-root $root $extra > $O/rehearse_$tag.json 2> $O/rehearse_$tag.err || { echo "rehearsal $cfg FAILED"; tail -15 $O/rehearse_$tag.err; }
  python - $O/rehearse_$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("rehearse", d["n_gpus"], "ranks ok=%s merged_ok=%s %.1f Mints/s :: %s"%(d["roundtrip_ok"],d["merged_container_ok"],d["value"],str(d["config"]["multi_gpu"]["root_policy"])))
except Exception as e: print("rehearsal output unreadable", e)
PY
done
