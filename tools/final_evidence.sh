#!/bin/bash
# Round-end evidence run on the GPU box, ONE per round:   gpurun --timeout 3000 -- 'bash tools/final_evidence.sh r06z'
# GPU tests, the driver's bench line and one line per BASELINE configuration, rocprofv3 kernel-trace summaries + PMC traffic of
# cfg2 / cfg3 / cfg4, of the long-filter plan and of two SWT plans, the long-filter and SWT A/Bs, the operator rates, the size sweep -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/ as <tag>_*).
set -uo pipefail
TAG="${1:-r06z}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1 || echo "pytest exit $?" >> "$OUT/pytest.log"
grep -E "passed|failed|error" "$OUT/pytest.log" | tail -2
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || echo "smoke exit $?"
timeout 400 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || echo "bench default exit $?"
for c in cfg1 cfg3 cfg4; do
    timeout 400 python3 bench.py --config $c --no-extras > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || echo "bench $c exit $?"
done
timeout 400 python3 bench.py --config cfg5 --no-extras > "$OUT/bench_cfg5_shard.json" 2> "$OUT/bench_cfg5.err" || echo "bench cfg5 exit $?"
timeout 400 python3 bench.py --config cfg2 --batch 16 --no-extras > "$OUT/bench_cfg2_b16.json" 2> "$OUT/bench_cfg2_b16.err" || echo "bench b16 exit $?"
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 300 python3 bench.py --force-dist --dist-backend nccl --no-extras > "$OUT/bench_nccl_world1.json" 2> "$OUT/bench_nccl_world1.err" || echo "nccl world1 exit $?"
timeout 600 python3 tools/long_ab.py > "$OUT/long_ab.txt" 2> /dev/null || echo "long_ab exit $?"
timeout 600 python3 tools/f64long_ab.py 2> /dev/null | grep -v "Warning\|Forcing" > "$OUT/f64_long_ab.txt" || echo "f64long exit $?"
timeout 300 python3 tools/opsbench.py 2> /dev/null | grep -v "Warning\|Forcing" > "$OUT/opsbench.txt" || echo "opsbench exit $?"
timeout 900 python3 tools/cliffs.py case dwt2:db13:4096x4096:3:1 dwt2:db16:4096x4096:3:1 dwt2:db20:4096x4096:3:1 dwt2:db13:2048x2048:3:1 dwt2:db16:2048x2048:3:1 dwt2:db20:2048x2048:3:1 dwt2:db20:2048x2048:5:1 dwt2:db20:4096x4096:3:16 > "$OUT/cliffs_long.txt" 2> /dev/null || echo "cliffs exit $?"
timeout 900 python3 tools/sizes_cliff.py > "$OUT/sizes_cliff.txt" 2> /dev/null || echo "sizes exit $?"
timeout 600 python3 tools/swt_round6_ab.py 2> /dev/null | grep -v "^Warning\|^Forcing" > "$OUT/swt_round6_ab.txt" || echo "swt ab exit $?"
timeout 300 python3 tools/swt_any_ab.py 2> /dev/null | grep -v "^Warning\|^Forcing" > "$OUT/swt_any_ab.txt" || echo "swt any exit $?"
timeout 300 python3 tools/dispatch_table.py > "$OUT/dispatch_table.md" 2> /dev/null || echo "dispatch table exit $?"
timeout 600 python3 tools/refbench.py > "$OUT/refbench.txt" 2> "$OUT/refbench.err" || echo "refbench exit $?"
# rocprofv3: kernel trace + counters of the long-filter plan (db20 4096^2 L3, one image and four)
for b in 1 4; do
    bash tools/planprof.sh "${TAG}_db20_b$b" db20 4096 4096 3 $b > "$OUT/planprof_db20_b$b.log" 2>&1 || echo "planprof b$b exit $?"
    cp "gpurun_out/planprof_${TAG}_db20_b$b/summary.txt" "$OUT/rocprofv3_summary_db20_4096_L3_b$b.txt" 2> /dev/null || true
    rm -rf "gpurun_out/planprof_${TAG}_db20_b$b"
done
# rocprofv3: kernel trace + counters of the round's SWT kernels (db4 2048^2 L4: one launch per level both ways; db20 2048^2 L5: the reference benchmark's case)
for w in db4:4 db20:5; do
    bash tools/planprof.sh "${TAG}_swt_${w%%:*}" ${w%%:*} 2048 2048 ${w##*:} 1 1 > "$OUT/planprof_swt_${w%%:*}.log" 2>&1 || echo "planprof swt $w exit $?"
    cp "gpurun_out/planprof_${TAG}_swt_${w%%:*}/summary.txt" "$OUT/rocprofv3_summary_swt_${w%%:*}_2048_L${w##*:}.txt" 2> /dev/null || true
    rm -rf "gpurun_out/planprof_${TAG}_swt_${w%%:*}"
done
for c in cfg2 cfg3 cfg4; do
    tools/prof.sh "${TAG}_$c" --config $c > "$OUT/prof_$c.log" 2>&1 || echo "prof $c exit $?"
    python3 tools/summarize_pmc.py "gpurun_out/prof_${TAG}_$c" "$OUT/traffic_$c.json" $c > "$OUT/rocprofv3_summary_$c.txt" 2>&1 || echo "summarize $c exit $?"
    cp "$(ls gpurun_out/prof_${TAG}_$c/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_$c.csv" || true
    rm -rf "gpurun_out/prof_${TAG}_$c"
done
echo done
