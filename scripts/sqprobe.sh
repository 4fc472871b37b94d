#!/bin/bash
# one-off SQ counter probe: scripts/sqprobe.sh <tag> [bench args]
TAG=${1:-probe}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c 'import torch; torch.zeros(1).cuda()' 2>/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS SQ_ACTIVE_INST_VALU2"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -- \
      python3 bench.py --steps 1 --warmup 0 --no-cpu --no-profile --no-extra "$@" > /dev/null 2> $OUT/sq$i.err || echo "SQ set $i failed"
done
python3 - $OUT <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+'/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in sorted(acc):
    a={c:acc[k][c]/cnt[k][c] for c in acc[k]}
    w=a.get('SQ_WAVES',1)
    g=lambda c:a.get(c,0.0)
    wc=g('SQ_WAVE_CYCLES') or 1.0
    print('%-34s waves %7d  cyc/wave %9.0f  insts/wave %8.0f (valu %7.0f lds %6.0f vmem %6.0f salu %6.0f)  active: any %.2f valu %.2f lds %.2f vmem %.2f  wait_inst %.2f wait_any %.2f  busy_cyc %.0f'%(
        k[:34], w, 4*wc/w, g('SQ_INSTS')/w, g('SQ_INSTS_VALU')/w, g('SQ_INSTS_LDS')/w, (g('SQ_INSTS_VMEM_WR')+g('SQ_INSTS_VMEM_RD'))/w, g('SQ_INSTS_SALU')/w,
        g('SQ_ACTIVE_INST_ANY')/wc, g('SQ_ACTIVE_INST_VALU')/wc, g('SQ_ACTIVE_INST_LDS')/wc, g('SQ_ACTIVE_INST_VMEM')/wc, g('SQ_WAIT_INST_ANY')/wc, g('SQ_WAIT_ANY')/wc, 4*g('SQ_BUSY_CYCLES')))
PY
