#!/bin/bash
for n in 33554432 16777216 8388608 4194304 2097152; do echo "== ints $n"; timeout -k 10 200 bash scripts/quick.sh n_$n --no-extra --ints $n | sed 's/fold_hist.*encode=/encode=/; s/scan_sizes.*parse/parse/'; done
