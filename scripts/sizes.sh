#!/bin/bash
for n in 268435456 201326592 134217728 67108864 33554432; do echo "== ints $n"; timeout -k 10 200 bash scripts/quick.sh n_$n --no-extra --ints $n | sed 's/fold_hist.*encode=/encode=/; s/scan_sizes.*parse/parse/'; done
