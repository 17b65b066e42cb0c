#!/bin/bash
out=gpurun_out/exp12; mkdir -p $out
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
FSKHIP_SPLIT=4 timeout 600 bash tools/pmc.sh r03blk_insts "$C" --seconds 1 --steps 3 --warmup 1 --no-side > $out/pmc_insts_blk.txt 2>&1
FSKHIP_SPLIT=1 timeout 600 bash tools/pmc.sh r03pipe_insts "$C" --seconds 1 --steps 3 --warmup 1 --no-side > $out/pmc_insts_pipe.txt 2>&1
C2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
FSKHIP_SPLIT=4 timeout 600 bash tools/pmc.sh r03blk_act "$C2" --seconds 1 --steps 3 --warmup 1 --no-side > $out/pmc_act_blk.txt 2>&1
rm -rf gpurun_out/pmc_r03*
