#!/bin/bash
# where the four waves' time goes now (y ring 16 deep), and what the per-sample slow path costs (ablation: skip it)
out=gpurun_out/experiment; mkdir -p $out
N=96000
{
for S in 8192 65536; do
echo "# $S x $N"
VAR_TIMEOUT=90 VAR_STAMPS=1 timeout 400 python tools/variants.py $S $N "stamp@stamp" "plain" "abl_none@abl:FSK_ABLATE=0" "abl_noslow@abl:FSK_ABLATE=8" "abl_noC@abl:FSK_ABLATE=12" "abl_noBC@abl:FSK_ABLATE=14" "abl_all@abl:FSK_ABLATE=15"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
