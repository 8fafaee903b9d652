#!/bin/bash
set -u
OUT=gpurun_out/long6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_long.py -x -q > $OUT/pytest_long.txt 2>&1; echo "pytest long rc=$?"
tail -15 $OUT/pytest_long.txt
timeout 600 python3 tools/long_ab.py > $OUT/long_ab.txt 2>&1
cat $OUT/long_ab.txt
