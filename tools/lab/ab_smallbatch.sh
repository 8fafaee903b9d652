#!/bin/bash
# batches of small images: the wave kernels on narrow images (PDWT_WAVE_MIN_NC=0 = rule of rounds 2-3) against the LDS tiles
cd "$(dirname "${BASH_SOURCE[0]}")/.."
C="dwt2:db4:256x256:3:256 dwt2:haar:256x256:3:256 dwt2:db2:256x256:3:256 dwt2:db4:256x256:3:2048 dwt2:db4:512x512:3:64 dwt2:db4:512x512:3:512 dwt2:haar:512x512:3:64 dwt2:db4:128x128:3:1024 dwt2:db4:128x128:3:256 dwt2:db4:256x256:4:128 dwt2:db4:256x256:3:64 dwt2:db4:1024x1024:3:16 dwt2:db4:512x512:4:16 dwt2:db4:256x512:3:128 dwt2:db4:512x256:3:128"
for env in "PDWT_WAVE_MIN_NC=0" "X=1" "PDWT_WAVE_MIN_NC=512" "PDWT_WAVE_MIN_NC=0" "X=1"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $C 2>&1 | grep -v "^#" | cut -c1-200
done
