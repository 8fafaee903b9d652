#!/bin/bash
set -u
OUT=gpurun_out/tune1
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
timeout 900 python3 tools/retune.py > $OUT/retune.txt 2>&1; echo "retune rc=$?"
cat $OUT/retune.txt
timeout 1200 python3 tools/sched_scan.py > $OUT/sched_scan.txt 2>&1; echo "sched_scan rc=$?"
grep -c "default loses" $OUT/sched_scan.txt; grep "default loses" $OUT/sched_scan.txt | head
