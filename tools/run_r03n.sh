mkdir -p gpurun_out/r03n
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wave.py tests/test_gpu_ops.py -x -q -k "swt or cfg4 or haar or threshold" > gpurun_out/r03n/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r03n/pytest.log | tail -2
timeout 300 python3 bench.py --config cfg4 --no-extras --no-cpu-baseline > gpurun_out/r03n/bench_cfg4.json 2> gpurun_out/r03n/bench_cfg4.err
export TMPDIR=/tmp
tools/prof.sh r03n_cfg4 --config cfg4 > gpurun_out/r03n/prof.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/prof_r03n_cfg4 gpurun_out/r03n/traffic_cfg4.json cfg4 > gpurun_out/r03n/summary_cfg4.txt 2>&1
rm -rf gpurun_out/prof_r03n_cfg4
