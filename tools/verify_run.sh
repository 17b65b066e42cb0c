#!/bin/bash
# usage (GPU box): tools/verify_run.sh -- smoke, the whole GPU suite, one default bench line
out=gpurun_out/verify; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
timeout 900 python bench.py > $out/bench.txt 2>&1; tail -c 400 $out/bench.txt
