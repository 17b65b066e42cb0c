#!/bin/bash
out=gpurun_out/exp4; mkdir -p $out
N=96000
for pat in "r3 "; do timeout 300 ./tools/build/valu_probe "$pat" > $out/valu_probe_r3.txt 2>&1; done
{
for S in 8192 65536; do
echo "# back-wave cuts (two-wave kernel), $S x $N"
timeout 900 python tools/variants.py $S $N "stamp@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut1_ampstore@cut1:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut2_poly_matched@cut2:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut4_raretest@cut4:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut8_edge@cut8:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut16_discr@cut16:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut32_vote@cut32:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut64_zlive@cut64:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut12_bothbranches@cut12:VAR_STAMPS=1,FSKHIP_SPLIT=1" "cut127_all@cut127:VAR_STAMPS=1,FSKHIP_SPLIT=1"
done
} > $out/variants.txt 2>&1
