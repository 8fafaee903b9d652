#!/bin/bash
# A/B of the all-remaining-levels launch (dwt2_tail_kernels.hpp) against the pyramid / level launches on deep plans:
# the dwt2 part of the reference's benchmark set and small-image cases, forward and forward+inverse, same box.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
C="dwt2:haar:32x32:5 dwt2:haar:64x64:6 dwt2:haar:128x128:7 dwt2:haar:256x256:8 dwt2:haar:512x512:9 dwt2:haar:1024x1024:10 dwt2:haar:2048x2048:11 dwt2:haar:4096x4096:12 dwt2:db2:128x128:5 dwt2:db2:256x256:6 dwt2:db2:512x512:7 dwt2:db2:1024x1024:8 dwt2:db2:2048x2048:9 dwt2:db4:256x256:5 dwt2:db4:512x512:6 dwt2:db4:2048x2048:8 dwt2:sym8:1024x1024:6 dwt2:sym8:512x512:5 dwt2:haar:128x128:7:16 dwt2:haar:128x128:7:64 dwt2:db2:256x256:6:16 dwt2:haar:64x256:6:4"
for env in "PDWT_NO_TAIL=1" "X=1" "PDWT_TAIL_WORK_LOG2=15 PDWT_TAIL_MIN_K=4" "PDWT_TAIL_WORK_LOG2=16 PDWT_TAIL_MIN_K=3" "PDWT_NO_TAIL=1" "X=1"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $C 2>&1 | grep -v "^#" | cut -c1-230
done
