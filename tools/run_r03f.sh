mkdir -p gpurun_out/r03f tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench 2> gpurun_out/r03f/qb_build.err
for cfg in "0 128" "0 144"; do set -- $cfg; echo "== lag $1 mode $2"; timeout 120 tools/bin/queuebench $1 4 4096 8 $2 | grep -v "alone"; done > gpurun_out/r03f/queuebench_cont.txt 2>&1
echo "== 2 levels, mode 144" >> gpurun_out/r03f/queuebench_cont.txt; timeout 120 tools/bin/queuebench 0 2 4096 8 144 | grep -v "alone\|epoch set" >> gpurun_out/r03f/queuebench_cont.txt 2>&1
echo done
