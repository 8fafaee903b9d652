#!/bin/bash
# same-box check of the round-4 batch-range rules against the rules of round 3 (PDWT_LDS_MAX=24 PDWT_STRIP_MIN_LOG2=26 PDWT_SWT_FUSED=1 is
# not the old SWT rule -- the old SWT rule cannot be selected any more; the SWT lines are before/after across boxes)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
D="dwt2:db4:4096x4096:4:2 dwt2:db4:4096x4096:4:3 dwt2:haar:4096x4096:4:2 dwt2:haar:4096x4096:4:3 dwt2:db2:4096x4096:4:2 dwt2:db4:2048x2048:4:8 dwt2:db4:2048x2048:4:12 dwt2:haar:2048x2048:4:8 dwt2:db4:1024x1024:3:32 dwt2:db4:1024x1024:3:48 dwt2:haar:1024x1024:3:32 dwt2:sym8:4096x4096:4:2"
for rep in 1 2; do
for env in "X=1" "PDWT_LDS_MAX=24 PDWT_STRIP_MIN_LOG2=26"; do
    echo "== $env"
    env $env python3 tools/cliffs.py case $D 2>&1 | grep -v "^#"
done
done
