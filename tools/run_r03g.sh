mkdir -p gpurun_out/r03g
timeout 900 python3 -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/r03g/pytest_chain.log 2>&1; tail -15 gpurun_out/r03g/pytest_chain.log
for c in 1 0; do PDWT_CHAIN=$c timeout 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r03g/bench_cfg2_chain$c.json 2> gpurun_out/r03g/bench_cfg2_chain$c.err; done
for c in 1 0 3; do PDWT_CHAIN=$c timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03g/bench_b16_chain$c.json 2> gpurun_out/r03g/bench_b16_chain$c.err; done
echo done
