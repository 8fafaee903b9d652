mkdir -p gpurun_out/r03r
timeout 1500 python3 -m pytest tests -q -m gpu > gpurun_out/r03r/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r03r/pytest.log | tail -2
timeout 300 python3 bench.py --no-extras > gpurun_out/r03r/bench_default.json 2> gpurun_out/r03r/bench_default.err
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03r/smoke.log 2>&1; tail -1 gpurun_out/r03r/smoke.log
