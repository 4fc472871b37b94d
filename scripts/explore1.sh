#!/bin/bash
# round-2 exploration: parity suite + bench at several geometries / configs (per-kernel times)
set -o pipefail
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python bench.py --steps 10 --warmup 2 > $O/b_default.json 2> $O/b_default.err || exit 1
for cfg in "--block 8192 --ckpt 1024" "--block 8192 --ckpt 2048" "--block 16384 --ckpt 2048" "--block 4096 --ckpt 1024" \
           "--fidelity 3 --dist zipf24s1.2" "--codec rfold --fidelity 3 --dist zipf24s1.2" "--fidelity 3 --dist zipf24s1.0" "--dist uniform256" "--codec rfold"; do
  tag=$(echo $cfg | tr -d ' -' )
  echo "== $cfg"
  timeout -k 10 180 python bench.py --steps 5 --warmup 1 --no-cpu $cfg > $O/b_$tag.json 2> $O/b_$tag.err || { echo FAILED; tail -3 $O/b_$tag.err; }
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/r2a/b_*.json')):
    try: d=json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: print(p,'unreadable'); continue
    k=d.get('kernels') or {}
    print(p.split('/')[-1], '%.1f Gints/s'%(d['value']/1e3), '%.3f ms'%d['ms_per_step'], 'bpi %.3f'%d['bits_per_int'], ' '.join('%s=%.3f'%(n.replace('k_',''),v['avg_ms']*v['launches_per_step']) for n,v in k.items()))
PY
