mkdir -p gpurun_out/r03b tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench 2> gpurun_out/r03b/qb_build.err
for lag in 0 4 8 16 32 48; do echo "== lag $lag"; timeout 120 tools/bin/queuebench $lag 4 4096 8; done > gpurun_out/r03b/queuebench.txt 2>&1
echo "== 2 levels" >> gpurun_out/r03b/queuebench.txt; timeout 120 tools/bin/queuebench 8 2 4096 8 >> gpurun_out/r03b/queuebench.txt 2>&1
timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03b/bench_cfg2_b16.json 2> gpurun_out/r03b/bench_cfg2_b16.err
timeout 300 python3 tools/torchcoexist.py > gpurun_out/r03b/torchcoexist.txt 2>&1
echo done
