#!/bin/bash
out=gpurun_out/exp33; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "y_ring or time_sliced" > $out/tests.txt 2>&1
tail -3 $out/tests.txt
timeout 1500 python tools/soak.py 1200 9917 > $out/soak_b.txt 2>&1
tail -3 $out/soak_b.txt
