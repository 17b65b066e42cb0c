#!/bin/bash
# usage (GPU box): tools/soak_run.sh <seconds> <seed>  -- the parity tests of the newest kernels, then tools/soak.py
out=gpurun_out/soak; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_bit_cells" > $out/tests.txt 2>&1
tail -3 $out/tests.txt
timeout $(( $1 + 300 )) python tools/soak.py $1 $2 > $out/soak_$2.txt 2>&1
tail -3 $out/soak_$2.txt
