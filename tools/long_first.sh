#!/bin/bash
set -u
OUT=gpurun_out/pkg1
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_packaging.py tests/test_cython_shim.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.txt
