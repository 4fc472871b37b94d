#!/bin/bash
# block size / restart interval sweep on the headline workload
for cfg in "16384 1024" "16384 2048" "16384 512" "8192 1024" "8192 2048" "32768 1024" "4096 1024"; do
  set -- $cfg
  echo "== block $1 ckpt $2"
  timeout -k 10 200 bash scripts/quick.sh s_$1_$2 --no-extra --block $1 --ckpt $2 | sed 's/fold_hist.*encode=/encode=/; s/scan_sizes.*parse/parse/'
done
