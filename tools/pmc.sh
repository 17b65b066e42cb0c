#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args...]   (run on the GPU box via gpurun)
# Collects PMC counters for the bench in their own rocprofv3 pass (no tracing domains with --pmc).
tag=$1; shift
ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in acc.items():
    print(k, "dispatches", len(n[k]))
    for c,x in v.items(): print("   %-28s total %.6g  per-dispatch %.6g" % (c, x, x/max(1,len(n[k]))))
PY
