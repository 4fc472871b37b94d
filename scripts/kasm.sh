#!/bin/bash
# scripts/kasm.sh <mangled-prefix> : regenerate the ISA listing and print one kernel's body to /tmp/kasm.s
cd "$(dirname "$(readlink -f "$0")")/../ans_large_alphabet_amd/csrc" && make asm >/dev/null 2>&1
grep -i " error" -A5 resource_usage.txt | head -20
s=$(grep -n "^$1.*:" ansx_gfx950.s | head -1 | cut -d: -f1)
awk -v s=$s 'NR>=s' ansx_gfx950.s | awk '/^\.Lfunc_end/{print; exit} {print}' > /tmp/kasm.s
wc -l /tmp/kasm.s; grep "NumVgprs\|ScratchSize" /tmp/kasm.s
