mkdir -p gpurun_out/r03p
for seg in 256 512 1024 2048 4096; do PDWT_INV_STRIP=1 PDWT_ISTRIP_SEG=$seg timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03p/b16_istrip_seg$seg.json 2>/dev/null; done
timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03p/b16_default.json 2>/dev/null
timeout 300 python3 tools/dispatch_table.py > gpurun_out/r03p/dispatch_table.md 2>/dev/null
