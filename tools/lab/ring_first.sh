#!/bin/bash
# first GPU run of the ring kernels: micro-benchmark, parity, per-launch times (developer tool)
set -u
OUT=gpurun_out/ring1
mkdir -p $OUT
tools/bin/pkbench > $OUT/pkbench.txt 2>&1
for cpl in 2 4; do
  PDWT_RING=$cpl PDWT_RING_MIN=10 timeout 600 python3 tools/ringcheck.py parity > $OUT/parity_cpl$cpl.txt 2>&1
  echo "parity cpl=$cpl rc=$?"
done
SPECS="sym8,4096,4096,4 db10,4096,4096,1 db5,4096,4096,1 db20,2048,2048,5"
python3 tools/ringcheck.py time $SPECS > $OUT/time_base.txt 2>&1
for cpl in 2 4; do
  for seg in 0 16 32 64; do
    PDWT_RING=$cpl PDWT_RING_SEG=$seg python3 tools/ringcheck.py time sym8,4096,4096,4 db10,4096,4096,1 db5,4096,4096,1 > $OUT/time_cpl${cpl}_seg$seg.txt 2>&1
  done
done
grep -h "pipelined\|PDWT_RING\|level\[L1\]\|dwt2_fwd_level \|dwt2_inv_level " $OUT/time_*.txt | head -150
tail -3 $OUT/parity_cpl*.txt
cat $OUT/pkbench.txt
