#!/bin/bash
# Round-end evidence run on the GPU box:   gpurun -- 'bash tools/final_evidence.sh r05z'
# GPU tests, every bench line, rocprofv3 kernel-trace summaries + PMC traffic per configuration -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/).
set -euo pipefail
TAG="${1:-r05z}"
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
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 300 python3 bench.py --force-dist --dist-backend nccl --no-extras > "$OUT/bench_nccl_world1.json" 2> "$OUT/bench_nccl_world1.err" || echo "nccl world1 exit $?"
timeout 900 python3 tools/refbench.py > "$OUT/refbench.txt" 2> "$OUT/refbench.err" || echo "refbench exit $?"
timeout 300 python3 tools/dispatch_table.py > "$OUT/dispatch_table.md" 2> /dev/null || echo "dispatch table exit $?"
timeout 300 python3 tools/oddtime.py > "$OUT/oddtime.txt" 2> /dev/null || echo "oddtime exit $?"
timeout 900 python3 tools/cliffs.py dwt2 swt2 dwt1 swt1 > "$OUT/cliffs.txt" 2> /dev/null || echo "cliffs exit $?"
timeout 300 python3 tools/opsbench.py > "$OUT/opsbench.txt" 2> /dev/null || echo "opsbench exit $?"
PDWT_BENCH_SHARE_GPU=1 timeout 400 python3 bench.py --gpus 2 --single-process --config cfg2 --batch 8 --no-cpu-baseline > "$OUT/bench_single_process_two_shards_one_gpu.json" 2> "$OUT/bench_single_process.err" || echo "single-process exit $?"
timeout 600 python3 tools/tiledbench.py > "$OUT/tiledbench.txt" 2> "$OUT/tiledbench.err" || echo "tiledbench exit $?"
timeout 900 python3 tools/f64scan.py 2> /dev/null | grep -v Warning > "$OUT/f64scan.txt" || echo "f64scan exit $?"
timeout 600 python3 tools/swt_stream32_ab.py 2> /dev/null | grep -v Warning > "$OUT/swt_stream32_ab.txt" || echo "stream32 exit $?"
# round 5: the register-ring level kernels next to the LDS tiles -- alone (same harness) and inside plans (same process)
for h in 10 12 14 16 18 20; do
    [ -x tools/bin/ringbench_${h}_4 ] && { tools/bin/ringbench_${h}_4 4096 $((h)) 1; tools/bin/ringbench_${h}_4 4096 $((2 * h)) 4; tools/bin/ringbench_${h}_4 2048 $((h / 2)) 1; } >> "$OUT/ringbench.txt" 2>&1 || true
done
timeout 600 python3 tools/ring_ab.py > "$OUT/ring_ab.txt" 2> /dev/null || echo "ring_ab exit $?"
timeout 300 python3 tools/dispatch_discover.py > "$OUT/dispatch_discover.txt" 2> /dev/null || echo "dispatch_discover exit $?"
timeout 600 python3 tools/cliffs.py odd > "$OUT/cliffs_odd.txt" 2> /dev/null || echo "cliffs odd exit $?"
export TMPDIR=/tmp
PDWT_PLANPROF_BATCH=4 bash tools/planprof.sh "${TAG}_sym8_b4" sym8 4096 4096 1 4 > "$OUT/planprof_sym8_b4.log" 2>&1 || echo "planprof sym8 b4 exit $?"
cp "gpurun_out/planprof_${TAG}_sym8_b4/summary.txt" "$OUT/planprof_sym8_L1_b4_default_dispatch.txt" 2> /dev/null || true
rm -rf "gpurun_out/planprof_${TAG}_sym8_b4"
tools/prof.sh "${TAG}_cfg2_b16" --config cfg2 --batch 16 > "$OUT/prof_cfg2_b16.log" 2>&1 || echo "prof b16 exit $?"
python3 tools/summarize_pmc.py "gpurun_out/prof_${TAG}_cfg2_b16" "$OUT/traffic_cfg2_b16.json" cfg2 > "$OUT/rocprofv3_summary_cfg2_b16.txt" 2>&1 || echo "summarize b16 exit $?"
rm -rf "gpurun_out/prof_${TAG}_cfg2_b16"
for c in cfg2 cfg3 cfg4; do
    tools/prof.sh "${TAG}_$c" --config $c > "$OUT/prof_$c.log" 2>&1 || echo "prof $c exit $?"
    python3 tools/summarize_pmc.py "gpurun_out/prof_${TAG}_$c" "$OUT/traffic_$c.json" $c > "$OUT/rocprofv3_summary_$c.txt" 2>&1 || echo "summarize $c exit $?"
    cp "$(ls gpurun_out/prof_${TAG}_$c/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_$c.csv" || true
    rm -rf "gpurun_out/prof_${TAG}_$c"
done
echo done
