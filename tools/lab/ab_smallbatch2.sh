#!/bin/bash
# batches of small images, the whole set of round-4 rules (narrow-image tiles, no wave kernels below 512 columns, tail launch in batch
# mode) against the rules of round 3 as far as they can still be selected: PDWT_WAVE_MIN_NC=0 PDWT_TAIL_BATCH=0 PDWT_NO_TAIL=1
cd "$(dirname "${BASH_SOURCE[0]}")/.."
C="dwt2:db4:256x256:3:256 dwt2:haar:256x256:3:256 dwt2:db4:256x256:3:2048 dwt2:db4:512x512:3:64 dwt2:db4:128x128:3:1024 dwt2:haar:128x128:3:1024 dwt2:db2:128x128:5:256 dwt2:haar:128x128:7:128 dwt2:db4:64x64:3:4096 dwt2:db4:64x64:3:256 dwt2:db2:32x32:3:4096 dwt2:sym8:64x64:2:1024 dwt2:db4:32x64:2:512 dwt2:haar:16x16:4:8192 dwt2:db4:1024x1024:3:16 dwt2:db4:4096x4096:4:1"
for env in "PDWT_WAVE_MIN_NC=0 PDWT_TAIL_BATCH=0 PDWT_NO_TAIL=1" "X=1" "PDWT_WAVE_MIN_NC=0 PDWT_TAIL_BATCH=0 PDWT_NO_TAIL=1" "X=1"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $C 2>&1 | grep -v "^#" | cut -c1-220
done
