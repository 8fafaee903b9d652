mkdir -p gpurun_out/dbg
for d in 1 2 3; do echo "== dbg $d"; PDWT_CHAIN_DBG=$d python3 tools/chaindbg.py db4 2>&1 | grep -v "max diff 0.0"; done > gpurun_out/dbg/chaindbg.txt 2>&1
