#!/bin/bash
set -u
OUT=gpurun_out/f64long
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_gpu.txt | head
