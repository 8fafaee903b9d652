#!/bin/bash
# A/B: one wavefront per image in the SWT tail launch (PDWT_TAIL_WAVE_MAX = largest image in samples; 0 = never)
C="swt2:haar:16x16:3:65536 swt2:haar:32x32:3:16384 swt2:db4:32x32:2:8192 swt2:db2:16x16:2:65536 swt2:haar:8x8:2:262144 swt2:db2:28x28:2:20000 swt2:haar:12x20:2:5000"
for m in 0 1024; do echo "== PDWT_TAIL_WAVE_MAX=$m"; PDWT_TAIL_WAVE_MAX=$m python tools/cliffs.py case $C 2>&1 | grep '^swt2'; done
