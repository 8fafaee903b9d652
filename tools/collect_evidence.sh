#!/bin/bash
# copies what tools/final_evidence.sh left in gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_evidence.sh r06z
set -euo pipefail
TAG="${1:-r06z}"
ROUND="${TAG:0:3}"
cd "$(dirname "${BASH_SOURCE[0]}")/.."
S="gpurun_out/$TAG"
for f in bench_default bench_cfg1 bench_cfg3 bench_cfg4 bench_cfg5_shard bench_cfg2_b16 bench_nccl_world1; do
    [ -s "$S/$f.json" ] && cp "$S/$f.json" "profiles/${TAG}_$f.json"
done
for f in rocprofv3_summary_cfg2 rocprofv3_summary_cfg3 rocprofv3_summary_cfg4 rocprofv3_summary_db20_4096_L3_b1 rocprofv3_summary_db20_4096_L3_b4 \
         long_ab f64_long_ab opsbench cliffs_long sizes_cliff refbench swt_round6_ab swt_any_ab rocprofv3_summary_swt_db4_2048_L4 rocprofv3_summary_swt_db20_2048_L5; do
    [ -s "$S/$f.txt" ] && cp "$S/$f.txt" "profiles/${TAG}_$f.txt"
done
[ -s "$S/dispatch_table.md" ] && cp "$S/dispatch_table.md" "profiles/${TAG}_dispatch_table.md"
[ -s "$S/smoke.log" ] && grep -v "^Warning\|Forcing nlevels" "$S/smoke.log" > "profiles/${TAG}_smoke.log"
for c in cfg2 cfg3 cfg4; do
    [ -s "$S/kernel_stats_$c.csv" ] && cp "$S/kernel_stats_$c.csv" "profiles/${TAG}_kernel_stats_$c.csv"
    [ -s "$S/traffic_$c.json" ] && cp "$S/traffic_$c.json" "profiles/${ROUND}_traffic_$c.json"
done
grep -v "^Warning\|Forcing nlevels" "$S/pytest.log" > "profiles/${TAG}_pytest_gpu.log"
sed -i '/^Warning: /d' profiles/${TAG}_refbench.txt profiles/${TAG}_cliffs_long.txt profiles/${TAG}_sizes_cliff.txt 2> /dev/null || true
ls profiles | grep -c "^${TAG}_"
