#!/bin/bash
# A/B: one wavefront per image in the tail launch (PDWT_TAIL_WAVE_MAX = largest image in samples; 0 = never)
C="dwt2:haar:16x16:4:65536 dwt2:db2:16x16:2:65536 dwt2:haar:8x8:3:262144 dwt2:db2:28x28:3:20000 dwt2:db2:32x32:3:16384 dwt2:db4:32x32:2:16384 dwt2:haar:12x20:2:5000 dwt2:db2:24x40:2:1200"
for m in 0 256 1024; do echo "== PDWT_TAIL_WAVE_MAX=$m"; PDWT_TAIL_WAVE_MAX=$m python tools/cliffs.py case $C 2>&1 | grep '^dwt2'; done
