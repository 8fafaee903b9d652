mkdir -p gpurun_out/r03e tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench 2> gpurun_out/r03e/qb_build.err
for cfg in "64 96" "96 96" "128 96" "192 96" "250 96" "96 33" "250 33"; do set -- $cfg; echo "== lag $1 mode $2"; timeout 120 tools/bin/queuebench $1 4 4096 8 $2 | grep -v "epoch set\|alone"; done > gpurun_out/r03e/queuebench_lag.txt 2>&1
echo done
