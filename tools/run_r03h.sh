mkdir -p gpurun_out/r03h
for x in 1 0; do for k in 2 3 4; do PDWT_CHAIN_XCD=$x PDWT_CHAIN_K=$k timeout 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r03h/bench_x${x}_k$k.json 2> gpurun_out/r03h/bench_x${x}_k$k.err; done; done
PDWT_CHAIN=0 timeout 300 python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r03h/bench_classic.json 2>/dev/null
echo done
