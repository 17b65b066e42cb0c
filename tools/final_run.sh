#!/bin/bash
# Round-end measurement pass (GPU box, through gpurun): suite, bench lines of every BASELINE configuration (each with its
# in-run oracle check: VERDICT r03 weak #8) and of the idle / never-syncing banks, rocprofv3 kernel trace of the bench
# command, PMC passes (instruction mix + busy cycles, HBM bytes), batch sweep.  Summaries land in gpurun_out/final/; copy
# what is to be judged into profiles/ (r06_*).
mkdir -p gpurun_out/final
F=gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q > $F/pytest.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k deviation > $F/deviation.txt 2>&1
timeout 900 python bench.py > $F/bench.txt 2>&1
timeout 600 python bench.py --workload c2 --streams 4096 --no-side > $F/bench_c2.txt 2>&1
timeout 600 python bench.py --workload c4 --streams 32768 --no-side > $F/bench_c4.txt 2>&1
timeout 900 python bench.py --workload c5 --streams 16384 --no-side > $F/bench_c5.txt 2>&1
timeout 600 python bench.py --workload idle --no-side > $F/bench_idle.txt 2>&1
timeout 600 python bench.py --workload idle4 --no-side --cpu-seconds 6 > $F/bench_idle4.txt 2>&1
timeout 600 python bench.py --workload mod > $F/bench_mod.txt 2>&1
timeout 600 python bench.py --workload mod --precision f64 > $F/bench_mod_f64.txt 2>&1
timeout 600 python bench.py --streams 2048 --no-side > $F/bench_2048.txt 2>&1
timeout 600 python bench.py --workload c1x --no-side > $F/bench_c1x.txt 2>&1
timeout 600 python bench.py --precision f64 --steps 2 --warmup 1 --no-side --cpu-seconds 6 > $F/bench_f64.txt 2>&1
timeout 600 python bench.py --streams 8192 --no-side > $F/bench_8192.txt 2>&1
timeout 600 python bench.py --streams 4096 --no-side > $F/bench_4096.txt 2>&1
timeout 600 python bench.py --streams 16384 --no-side > $F/bench_16384.txt 2>&1
timeout 600 python bench.py --lead-max 40000 --no-side > $F/bench_staggered.txt 2>&1
timeout 600 python bench.py --workload c4 --streams 4096 --no-side --cpu-seconds 6 > $F/bench_c4_4096.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$F/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-side --no-clock-probe > $R/$F/stats.log 2>&1
cd $R
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
timeout 400 bash tools/pmc.sh r06f_insts "$C" --seconds 1 --steps 3 --warmup 2 --no-side --no-clock-probe > $F/pmc_insts.txt 2>&1
timeout 400 bash tools/pmc.sh r06f_fetch "FETCH_SIZE" --seconds 1 --steps 3 --warmup 2 --no-side --no-clock-probe > $F/pmc_fetch.txt 2>&1
timeout 400 bash tools/pmc.sh r06f_write "WRITE_SIZE" --seconds 1 --steps 3 --warmup 2 --no-side --no-clock-probe > $F/pmc_write.txt 2>&1
timeout 400 bash tools/pmc.sh r06f_clock "GRBM_GUI_ACTIVE" --seconds 10 --steps 3 --warmup 1 --no-side > $F/pmc_clock.txt 2>&1
find $F/stats -name "*kernel_stats.csv" -exec cp {} $F/kernel_stats.csv \;
rm -rf $F/stats gpurun_out/pmc_r06f_*/
[ "$1" = "sweep" ] && timeout 1500 bash tools/batch_sweep.sh > $F/batch_sweep.txt 2>&1
true
