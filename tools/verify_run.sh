#!/bin/bash
# usage (GPU box): tools/verify_run.sh [outdir] -- smoke, the whole GPU suite, the Node GPU tests when the box has node, one default bench line
out=${1:-gpurun_out/verify}; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu -rs > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
if which node > /dev/null 2>&1; then
  { echo "node: $(which node) $(node --version)"; timeout 600 node tests/js/fsk_core_test.js gpu; echo "rc=$?"; timeout 600 node tests/js/next_rows_test.js gpu; echo "rc=$?"; } > $out/node_gpu.txt 2>&1
else
  echo "no node on this box" > $out/node_gpu.txt
fi
cat $out/node_gpu.txt
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-600 $out/bench.json
