#!/bin/bash
# Collect the evidence bench.py's roofline object cites.  Run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r01_c'
# Three separate rocprofv3 runs (kernel-trace stats, then one PMC pass per counter — never
# combined with other trace domains), outputs under gpurun_out/<tag>/ ; copy the summaries
# printed at the end into profiles/.
set -e -o pipefail
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( while true; do echo "[collect $TAG] $(date +%T)"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
python3 -c 'import torch; torch.zeros(1).cuda()' 2>/dev/null   # page the image in
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile > $OUT/fetch.json 2> $OUT/fetch.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile > $OUT/write.json 2> $OUT/write.err
python3 profiles/summarize.py $OUT $TAG
