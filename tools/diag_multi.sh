echo "#### single call"; timeout 120 python tools/diag_replay.py tools/diag_data/soak_r02_default_s48.npy '{}' 21248 64 2>&1 | grep -A3 "== fused" | cut -c1-260
sch="1168"; for i in $(seq 1 60); do sch="$sch,16"; done
echo "#### 16-sample calls after 1168"; timeout 200 python tools/diag_replay.py tools/diag_data/soak_r02_default_s48.npy '{}' $sch 64 2>&1 | grep -A70 "== fused" | grep -B3 -A4 "DIFF" | head -24 | cut -c1-260
