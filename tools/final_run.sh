mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k deviation > gpurun_out/final/deviation.txt 2>&1
timeout 900 python bench.py > gpurun_out/final/bench.txt 2>&1
timeout 600 python bench.py --workload c2 --streams 4096 --no-side > gpurun_out/final/bench_c2.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-side > $R/gpurun_out/final/stats.log 2>&1
cd $R
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
timeout 400 bash tools/pmc.sh r02f_insts "$C" --seconds 1 --steps 3 --warmup 2 --no-side > gpurun_out/final/pmc_insts.txt 2>&1
timeout 400 bash tools/pmc.sh r02f_fetch "FETCH_SIZE" --seconds 1 --steps 3 --warmup 2 --no-side > gpurun_out/final/pmc_fetch.txt 2>&1
timeout 400 bash tools/pmc.sh r02f_write "WRITE_SIZE" --seconds 1 --steps 3 --warmup 2 --no-side > gpurun_out/final/pmc_write.txt 2>&1
find gpurun_out/final/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
rm -rf gpurun_out/final/stats gpurun_out/pmc_r02f_*/
