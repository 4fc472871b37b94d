#!/bin/bash
# scripts/kprobe.sh <kernel-substring> [one_encode args]: per-variant average duration of one kernel (rocprofv3 stats)
set -o pipefail
K=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$(readlink -f "$0")")/.."
cp ans_large_alphabet_amd/libansx.so /tmp/libansx_orig.so
trap 'cp /tmp/libansx_orig.so ans_large_alphabet_amd/libansx.so' EXIT
for v in /tmp/libansx_orig.so variants/libansx_*.so; do
  tag=$(basename $v .so | sed 's/libansx_//')
  [ $v != /tmp/libansx_orig.so ] && cp $v ans_large_alphabet_amd/libansx.so
  rm -rf /tmp/kp_$tag
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$tag -- python3 tests/tools/one_encode.py "$@" > /tmp/kp_$tag.log 2>&1 || echo "run failed $tag"
  python3 - "$tag" "$K" $(find /tmp/kp_$tag -name '*kernel_stats.csv') <<'PY'
import csv, sys
tag, k = sys.argv[1], sys.argv[2]
for f in sys.argv[3:]:
    for r in csv.DictReader(open(f)):
        if k in r["Name"]: print("== %-12s %-40s calls %s avg %.1f us min %.1f max %.1f" % (tag, r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  tail -2 /tmp/kp_$tag.log | grep -v "^done" | cut -c1-200
done
