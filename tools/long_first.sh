#!/bin/bash
set -u
OUT=gpurun_out/ops1
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_long.py tests/test_gpu_dispatch.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.txt
timeout 300 python3 tools/opsbench.py > $OUT/opsbench.txt 2>&1
cat $OUT/opsbench.txt
timeout 600 python3 tools/long_ab.py db20:4096x4096:3:1 db14:4096x4096:3:1 db13:4096x4096:3:1 db10:4096x4096:3:1 db9:4096x4096:3:4 db20:2048x2048:5:1 db16:2048x2048:5:1 db15:2048x2048:5:1 db20:4096x4096:3:16 > $OUT/long_ab.txt 2>&1
cat $OUT/long_ab.txt
