#!/bin/bash
out=gpurun_out/experiment; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_soak_regressions.py tests/test_gpu_next.py -x -q -m gpu > $out/tests_full.txt 2>&1
tail -3 $out/tests_full.txt
bash tools/experiments/processor_quanta.sh > /dev/null 2>&1
head -5 $out/processor.txt | cut -c1-200
