mkdir -p gpurun_out/r03i
export TMPDIR=/tmp
export PDWT_CHAIN_K=2
tools/prof.sh r03i_chain2 --config cfg2 > gpurun_out/r03i/prof.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/prof_r03i_chain2 gpurun_out/r03i/traffic.json cfg2 > gpurun_out/r03i/summary_chain2.txt 2>&1
rm -rf gpurun_out/prof_r03i_chain2
echo done
