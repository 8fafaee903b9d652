#!/bin/bash
set -u
OUT=gpurun_out/ring2
mkdir -p $OUT
for cpl in 2 4; do
  PDWT_RING=$cpl PDWT_RING_MIN=10 timeout 600 python3 tools/ringcheck.py parity > $OUT/parity_cpl$cpl.txt 2>&1
  echo "parity cpl=$cpl rc=$?"
done
python3 tools/ringcheck.py time sym8,4096,4096,1 db10,4096,4096,1 db5,4096,4096,1 sym8,2048,2048,1 sym8,4096,4096,1,2 > $OUT/time_base.txt 2>&1
for cpl in 4 2; do
  for seg in 8 16 24 32; do
    PDWT_RING=$cpl PDWT_RING_SEG=$seg PDWT_RING_MIN=20 python3 tools/ringcheck.py time sym8,4096,4096,1 db10,4096,4096,1 db5,4096,4096,1 sym8,2048,2048,1 sym8,4096,4096,1,2 > $OUT/time_cpl${cpl}_seg$seg.txt 2>&1
  done
done
grep -h "pipelined\|PDWT_RING\|level " $OUT/time_*.txt
grep -h "parity:" $OUT/parity_cpl*.txt
PDWT_RING=4 PDWT_RING_SEG=16 bash tools/planprof.sh ring4b_sym8 sym8 4096 4096 1 > $OUT/pp_ring4b.txt 2>&1
grep -v "^$" gpurun_out/planprof_ring4b_sym8/summary.txt | grep "ring_kernel" | cut -c1-330
