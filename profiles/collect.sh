#!/bin/bash
# Collect the evidence bench.py's roofline object cites.  Run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r02_a'
#   gpurun --timeout 1100 -- 'BENCH_ARGS="--codec fold --fidelity 3 --dist zipf24s1.2" bash profiles/collect.sh r03_cfg3a'
# BENCH_ARGS selects the profiled configuration (default: bench.py's own = BASELINE config 2); the workload
# string in <tag>_hbm_traffic_pmc.json is what bench.py's roofline.traffic matches on.
# Separate rocprofv3 runs: kernel-trace stats, one PMC pass per HBM counter, and (SQ=1) three SQ
# counter sets -- counters are never combined with other trace domains.  Outputs under
# gpurun_out/<tag>/ ; copy the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r03_x}
BENCH_ARGS=${BENCH_ARGS:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( while true; do echo "[collect $TAG] $(date +%T)"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
python3 -c 'import torch; torch.zeros(1).cuda()' 2>/dev/null   # page the image in
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 bench.py --steps 10 --warmup 2 --no-extra --no-host-api $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile --no-extra $BENCH_ARGS > $OUT/fetch.json 2> $OUT/fetch.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile --no-extra $BENCH_ARGS > $OUT/write.json 2> $OUT/write.err
python3 profiles/summarize.py $OUT $TAG
if [ "${SQ:-1}" = "1" ]; then
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile --no-extra $BENCH_ARGS > /dev/null 2> $OUT/sq$i.err || echo "SQ set $i failed"
  done
  python3 profiles/summarize_sq.py $OUT $TAG || true
fi
