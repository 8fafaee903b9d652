#!/bin/bash
# A/B of the dispatch rules in the batch ranges the cliff hunt flagged (profiles/r04r_cliffs_batch_odd.txt):
#  (a) 2D DWT levels of 2^24 < samples < 2^26 (wave kernels) against the LDS tiles (PDWT_LDS_MAX=26) and strips
#  (b) batches of 2-tap / 4-tap SWT images beyond the Infinity Cache: a launch per level against the fused groups (PDWT_SWT_FUSED=2)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
D="dwt2:db4:4096x4096:4:2 dwt2:db4:4096x4096:4:3 dwt2:haar:4096x4096:4:2 dwt2:db2:4096x4096:4:2 dwt2:db4:2048x2048:4:8 dwt2:db4:2048x2048:4:12 dwt2:db4:1024x1024:3:32 dwt2:db4:1024x1024:3:48 dwt2:sym8:4096x4096:4:2"
for env in "X=1" "PDWT_LDS_MAX=26" "PDWT_NO_WAVE=1" "PDWT_FORCE_STRIP=1"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $D 2>&1 | grep -v "^#"
done
S="swt2:haar:2048x2048:3:2 swt2:haar:2048x2048:3:4 swt2:haar:2048x2048:5:2 swt2:haar:2048x2048:5:4 swt2:db2:2048x2048:3:2 swt2:db2:2048x2048:3:4 swt2:haar:512x512:3:64 swt2:db2:512x512:3:64 swt2:haar:1024x1024:3:16"
for env in "X=1" "PDWT_SWT_FUSED=2"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $S 2>&1 | grep -v "^#"
done
