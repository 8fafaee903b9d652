mkdir -p gpurun_out/r03d tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench 2> gpurun_out/r03d/qb_build.err
for cfg in "8 32" "0 32" "16 32" "32 32" "48 32" "8 96" "32 96" "48 96"; do set -- $cfg; echo "== lag $1 mode $2"; timeout 120 tools/bin/queuebench $1 4 4096 8 $2 | grep -v "epoch set\|alone"; done > gpurun_out/r03d/queuebench_flags.txt 2>&1
echo done
