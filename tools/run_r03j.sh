mkdir -p gpurun_out/r03j
PDWT_CHAIN=0 PDWT_FORCE_STRIP=1 timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03j/b16_forcestrip.json 2> gpurun_out/r03j/b16_forcestrip.err
PDWT_CHAIN=1 PDWT_CHAIN_K=2 timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03j/b16_chaink2.json 2> gpurun_out/r03j/b16_chaink2.err
PDWT_CHAIN=0 timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03j/b16_classic.json 2> gpurun_out/r03j/b16_classic.err
echo done
