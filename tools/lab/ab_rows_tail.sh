#!/bin/bash
# A/B: batched 1D transforms of short rows, several rows per one-wavefront workgroup (PDWT_ROWS_TAIL_ROW = longest such row, 0 = off;
# PDWT_ROWS_TAIL_SAMPLES = samples per workgroup)
C="dwt1:haar:65536x64:3 dwt1:db4:65536x64:3 dwt1:haar:262144x64:3 dwt1:db2:131072x32:2 dwt1:db2:32768x128:4 dwt1:sym8:16384x128:3 dwt1:db10:16384x128:2 dwt1:haar:16384x256:5 dwt1:db4:16384x256:3 dwt1:sym8:8192x256:4 dwt1:haar:8192x512:4 dwt1:sym8:8192x512:4 dwt1:db4:4096x256:5 dwt1:db2:2048x1024:5"
echo "== off"; PDWT_ROWS_TAIL_ROW=0 python tools/cliffs.py case $C 2>&1 | grep '^dwt1' | cut -c1-170
echo "== rows <= 1024, 1024 samples per workgroup"; PDWT_ROWS_TAIL_ROW=1024 python tools/cliffs.py case $C 2>&1 | grep '^dwt1' | cut -c1-170
