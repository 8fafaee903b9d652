#!/bin/bash
# Round-end evidence run on the GPU box:   gpurun -- 'bash tools/final_evidence.sh r03z'
# GPU tests, every bench line, rocprofv3 kernel-trace summaries + PMC traffic per configuration -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/).
set -euo pipefail
TAG="${1:-r03z}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
timeout 1500 python3 -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1 || echo "pytest exit $?" >> "$OUT/pytest.log"
grep -E "passed|failed|error" "$OUT/pytest.log" | tail -2
for c in cfg2 cfg1 cfg3 cfg4; do
    timeout 400 python3 bench.py --config $c > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || echo "bench $c exit $?"
done
timeout 400 python3 bench.py --config cfg5 --no-extras > "$OUT/bench_cfg5_shard.json" 2> "$OUT/bench_cfg5.err" || echo "bench cfg5 exit $?"
timeout 400 python3 bench.py --config cfg2 --batch 16 --no-extras > "$OUT/bench_cfg2_b16.json" 2> "$OUT/bench_cfg2_b16.err" || echo "bench b16 exit $?"
timeout 300 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || echo "bench default exit $?"
export TMPDIR=/tmp
for c in cfg2 cfg3 cfg4; do
    tools/prof.sh "${TAG}_$c" --config $c > "$OUT/prof_$c.log" 2>&1 || echo "prof $c exit $?"
    python3 tools/summarize_pmc.py "gpurun_out/prof_${TAG}_$c" "$OUT/traffic_$c.json" $c > "$OUT/rocprofv3_summary_$c.txt" 2>&1 || echo "summarize $c exit $?"
    cp "$(ls gpurun_out/prof_${TAG}_$c/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_$c.csv" || true
done
echo done
