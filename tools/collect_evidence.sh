#!/bin/bash
# copies what tools/final_evidence.sh left in gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_evidence.sh r05z
set -euo pipefail
TAG="${1:-r05z}"
cd "$(dirname "${BASH_SOURCE[0]}")/.."
S="gpurun_out/$TAG"
for f in bench_default bench_cfg1 bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg5_shard bench_cfg2_b16 bench_nccl_world1 bench_single_process_two_shards_one_gpu; do
    [ -s "$S/$f.json" ] && cp "$S/$f.json" "profiles/${TAG}_$f.json"
done
for f in rocprofv3_summary_cfg2 rocprofv3_summary_cfg3 rocprofv3_summary_cfg4 rocprofv3_summary_cfg2_b16 ringbench ring_ab dispatch_discover cliffs cliffs_odd refbench \
         planprof_sym8_L1_b4_default_dispatch oddtime opsbench tiledbench f64scan swt_stream32_ab; do
    [ -s "$S/$f.txt" ] && cp "$S/$f.txt" "profiles/${TAG}_$f.txt"
done
[ -s "$S/dispatch_table.md" ] && cp "$S/dispatch_table.md" "profiles/${TAG}_dispatch_table.md"
for c in cfg2 cfg3 cfg4; do
    [ -s "$S/kernel_stats_$c.csv" ] && cp "$S/kernel_stats_$c.csv" "profiles/${TAG}_kernel_stats_$c.csv"
done
for c in cfg2 cfg3 cfg4 cfg2_b16; do
    [ -s "$S/traffic_$c.json" ] && cp "$S/traffic_$c.json" "profiles/r05_traffic_$c.json"
done
grep -v "^Warning\|Forcing nlevels" "$S/pytest.log" > "profiles/${TAG}_pytest_gpu.log"
sed -i '/^Warning: /d' profiles/${TAG}_refbench.txt profiles/${TAG}_cliffs.txt profiles/${TAG}_cliffs_odd.txt 2> /dev/null || true
ls profiles | grep -c "^${TAG}_"
