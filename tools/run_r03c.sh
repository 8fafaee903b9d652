mkdir -p gpurun_out/r03c tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench 2> gpurun_out/r03c/qb_build.err
for mode in 0 16 1 3 7 15 2 4 8; do echo "== mode $mode"; timeout 120 tools/bin/queuebench 8 4 4096 8 $mode | grep -v "epoch set"; done > gpurun_out/r03c/queuebench_modes.txt 2>&1
timeout 300 python3 tools/distgap.py > gpurun_out/r03c/distgap.txt 2>&1
echo done
