#!/bin/bash
# second batch of tools/valu_probe rows (see run_probe.sh)
out=${1:-gpurun_out/probe2}
mkdir -p "$out"
: > "$out/valu_probe2.txt"
for pat in "v_lshlrev" "v_lshrrev" "v_ashrrev" "v_xor" "v_or_b32" "v_sub_f32" "v_add_f32" "v_mov_b32 VGPR" "v_and_or" "v_add3" "v_lshl_add" "v_alignbit" "v_min_f32" "v_sub_u32" "v_subrev" "v_mul_lo" "v_mul_u32_u24" "v_mad_u32" "v_bfe" "v_perm" "v_mul_legacy" "v_ldexp" "v_floor" "v_fract" "v_fmac" "clamp" "v_max_u32" "v_cmp_class" "v_sad" "v_dot2c"; do
  timeout 40 ./tools/build/valu_probe "$pat" >> "$out/valu_probe2.txt" 2>&1 || echo "# pattern '$pat': exit $?" >> "$out/valu_probe2.txt"
done
