#!/bin/bash
# A/B of library builds on the GPU box: scripts/variants.sh [bench args]   (every variants/libansx_*.so in turn)
set -o pipefail
cd "$(dirname "$(readlink -f "$0")")/.."
cp ans_large_alphabet_amd/libansx.so /tmp/libansx_orig.so
trap 'cp /tmp/libansx_orig.so ans_large_alphabet_amd/libansx.so' EXIT   # (an interrupted run must not leave a variant installed)
for v in variants/libansx_*.so; do
  tag=$(basename $v .so | sed 's/libansx_//')
  cp $v ans_large_alphabet_amd/libansx.so
  echo "== $tag"
  timeout -k 10 200 bash scripts/quick.sh v_$tag --no-extra "$@" || echo "FAILED $tag"
done
