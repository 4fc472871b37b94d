#!/bin/bash
for d in "$@"; do echo "== $d"; timeout -k 10 200 bash scripts/quick.sh d_$d --no-extra --dist $d | sed 's/fold_hist.*encode=/encode=/; s/scan_sizes.*decode=/decode=/'; done
