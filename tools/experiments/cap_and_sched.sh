#!/bin/bash
# (GPU box) round 6, late: the bound on hand-off waits (csrc/fsk_wait.h) works and what it costs.
# builds needed (tools/build_variant.sh): cap0 (-DFSK_SPIN_CAP_LOG2=0), head (the library before the bound: copied from a build of the
# parent commit)
cd "$(dirname "$0")/../.."
out=gpurun_out/cap_and_sched.txt
: > $out
echo "== hand-off bound at 1 poll: every launch ends early, flagged; nothing hangs" >> $out
timeout 1500 python tools/handoff_check.py cap0 >> $out 2>&1
echo "== config #3, 65 536 x 480 000 (four waves)" >> $out
VAR_TIMEOUT=200 python tools/variants.py 65536 480000 before@head bounded before-again@head bounded-again >> $out 2>&1
echo "== 8 192 x 480 000 (seven waves)" >> $out
VAR_TIMEOUT=200 python tools/variants.py 8192 480000 before@head bounded before-again@head bounded-again >> $out 2>&1
echo "== 2 048 x 480 000 (seven waves, narrow groups)" >> $out
VAR_TIMEOUT=200 python tools/variants.py 2048 480000 before@head bounded >> $out 2>&1
echo "== staggered (lead-ins up to 40 000), 65 536 x 480 000" >> $out
VAR_TIMEOUT=200 VAR_LEAD=40000 python tools/variants.py 65536 480000 before@head bounded >> $out 2>&1
echo "== idle bank" >> $out
VAR_TIMEOUT=200 VAR_WORKLOAD=idle python tools/variants.py 65536 480000 before@head bounded >> $out 2>&1
echo "== 16 384 x 480 000 (seven waves, whole-wave groups)" >> $out
VAR_TIMEOUT=200 python tools/variants.py 16384 480000 before@head bounded >> $out 2>&1
cat $out
