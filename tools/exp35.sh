#!/bin/bash
out=gpurun_out/exp35; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" > $out/smoke.txt 2>&1
tail -3 $out/smoke.txt
