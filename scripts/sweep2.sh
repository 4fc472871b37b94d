#!/bin/bash
for cfg in "20480 1024" "24576 1024" "28672 1024" "32768 1024" "49152 1024" "24576 512"; do
  set -- $cfg
  echo "== block $1 ckpt $2"
  timeout -k 10 200 bash scripts/quick.sh s_$1_$2 --no-extra --block $1 --ckpt $2 | sed 's/sort_entropy.*encode=/encode=/; s/scan_sizes.*parse/parse/'
done
