#!/bin/bash
# runs tools/build/valu_probe row group by row group, each under its own timeout (a probe kernel that never ends must
# not take the box with it); output -> gpurun_out/<dir>/valu_probe.txt
out=${1:-gpurun_out/probe}
mkdir -p "$out"
: > "$out/valu_probe.txt"
for pat in "v_fma" "v_mul" "v_fmaak" "v_med3" "v_pk" "v_rcp" "v_sqrt" "v_rsq" "v_sin" "AGC" "v_add_u32" "v_bcnt" "v_cndmask" "v_cmp" "v_mov" "v_max" "v_rndne" "v_bfi" "v_lshl" "v_and" "v_bitop3" "v_addc" "v_cvt" "s_nop" "s_add" "s_waitcnt" "ds_"; do
  timeout 60 ./tools/build/valu_probe "$pat" >> "$out/valu_probe.txt" 2>&1 || echo "# pattern '$pat': exit $?" >> "$out/valu_probe.txt"
done
