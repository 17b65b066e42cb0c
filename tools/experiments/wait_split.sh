#!/bin/bash
# counter-based account of the SIMDs' idle third under the headline launch (GPU box, via gpurun).
# Separate --pmc passes (no trace domains), the program directly after `--` (tools/pmc.sh).  65 536 streams x 48 000 samples
# per launch.  Output: gpurun_out/wait_split/*.txt -> tools/wait_split.py -> profiles/r06_wait_split.txt
O=gpurun_out/wait_split; mkdir -p $O
A="--seconds 1 --steps 3 --warmup 2 --no-side --no-clock-probe"
timeout 400 bash tools/pmc.sh ws_a "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" $A > $O/a.txt 2>&1
timeout 400 bash tools/pmc.sh ws_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" $A > $O/b.txt 2>&1
timeout 400 bash tools/pmc.sh ws_c "SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" $A > $O/c.txt 2>&1
timeout 400 bash tools/pmc.sh ws_d "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $A > $O/d.txt 2>&1
timeout 400 bash tools/pmc.sh ws_e "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH" $A > $O/e.txt 2>&1
timeout 400 bash tools/pmc.sh ws_f "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" $A > $O/f.txt 2>&1
for t in a b c d e f; do tail -3 gpurun_out/pmc_ws_$t.log > $O/$t.log 2>/dev/null; done
rm -rf gpurun_out/pmc_ws_*/
true
