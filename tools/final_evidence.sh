mkdir -p gpurun_out/r02z
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r02z/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r02z/pytest.log | tail -2
for c in cfg2 cfg1 cfg3 cfg4; do timeout 400 python bench.py --config $c > gpurun_out/r02z/bench_$c.json 2> gpurun_out/r02z/bench_$c.err; done
timeout 400 python bench.py --config cfg5 --no-extras > gpurun_out/r02z/bench_cfg5_shard.json 2> gpurun_out/r02z/bench_cfg5.err
timeout 400 python bench.py --config cfg2 --batch 16 --no-extras > gpurun_out/r02z/bench_cfg2_b16.json 2>/dev/null
timeout 300 python bench.py > gpurun_out/r02z/bench_default.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in cfg2 cfg3 cfg4; do tools/prof.sh r02z_$c --config $c > /dev/null 2>&1; python3 tools/summarize_pmc.py gpurun_out/prof_r02z_$c gpurun_out/r02z/r02_traffic_$c.json $c > gpurun_out/r02z/rocprofv3_summary_$c.txt 2>&1; cp $(ls gpurun_out/prof_r02z_$c/stats/*/*kernel_stats.csv | head -1) gpurun_out/r02z/kernel_stats_$c.csv; done
echo done
