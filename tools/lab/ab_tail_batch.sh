#!/bin/bash
# A/B: largest image the tail launch takes in batch mode (PDWT_TAIL_BATCH samples; default 4096, 16384 with five levels and more)
C="dwt2:db4:128x128:3:1024 dwt2:db2:128x128:3:256 dwt2:haar:128x128:3:1024 dwt2:db4:100x100:3:120 dwt2:db4:100x100:2:2000 dwt2:db2:96x96:3:500 dwt2:sym8:128x128:2:512 dwt2:db2:80x120:4:300 dwt2:db2:128x128:4:4096"
for m in 4096 16384; do echo "== PDWT_TAIL_BATCH=$m"; PDWT_TAIL_BATCH=$m python tools/cliffs.py case $C 2>&1 | grep '^dwt2'; done
