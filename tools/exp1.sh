#!/bin/bash
# round-3 experiment 1: three waves per group at four groups per CU; stage ablations at 65536 streams
out=gpurun_out/exp1; mkdir -p $out
timeout 600 python __graft_entry__.py smoke > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt
N=96000
{
echo "# 65536 x $N"
timeout 1500 python tools/variants.py 65536 $N "pipe" "pipe_s3@s3" "pipe3_s4:FSKHIP_SPLIT=3" "pipe3_s3@s3:FSKHIP_SPLIT=3" "fused:FSKHIP_SPLIT=0"
for S in 4096 8192 16384 32768 49152; do
echo "# $S x $N"
timeout 600 python tools/variants.py $S $N "pipe:FSKHIP_SPLIT=1" "pipe3_s3@s3:FSKHIP_SPLIT=3"
done
echo "# ablation, two-wave kernel, 65536 x $N (bit w = wave w skips its arithmetic)"
timeout 900 python tools/variants.py 65536 $N "abl0@abl:FSK_ABLATE=0" "abl_front_only@abl:FSK_ABLATE=2" "abl_back_only@abl:FSK_ABLATE=1" "abl_skeleton@abl:FSK_ABLATE=3"
echo "# ablation, three-wave kernel (3 slots), 65536 x $N"
timeout 1500 python tools/variants.py 65536 $N "abl0@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=0" "w0_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=6" "w1_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=5" "w2_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=3" "skeleton@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=7" "w01@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=4" "w12@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=1"
} > $out/variants.txt 2>&1
