#!/bin/bash
# (GPU box) round 6, late: the bounded polls' wait loops out of line again (__builtin_expect on the guards) -- against the parent
# library ("before"), and the bounded library without the hint ("in-line").  builds: head, asm (copies of those libraries)
cd "$(dirname "$0")/../.."
out=gpurun_out/cap_expect.txt
: > $out
for shape in "65536 480000" "8192 480000" "4096 480000"; do
  echo "== $shape" >> $out
  VAR_TIMEOUT=200 python tools/variants.py $shape before@head in-line@asm out-of-line before-again@head in-line-again@asm out-of-line-again >> $out 2>&1
done
echo "== staggered 65536 480000" >> $out
VAR_TIMEOUT=200 VAR_LEAD=40000 python tools/variants.py 65536 480000 before@head in-line@asm out-of-line before-again@head out-of-line-again >> $out 2>&1
echo "== idle 65536 480000" >> $out
VAR_TIMEOUT=200 VAR_WORKLOAD=idle python tools/variants.py 65536 480000 before@head in-line@asm out-of-line >> $out 2>&1
cat $out
