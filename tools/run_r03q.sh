mkdir -p gpurun_out/r03q
PDWT_INV_PYR_L1=1 timeout 300 python3 bench.py --config cfg2 --batch 16 --no-extras --no-cpu-baseline > gpurun_out/r03q/b16_invpyr.json 2>gpurun_out/r03q/b16_invpyr.err
PDWT_INV_PYR_L1=1 timeout 300 python3 bench.py --config cfg2 --no-extras --no-cpu-baseline > gpurun_out/r03q/b1_invpyr.json 2>/dev/null
