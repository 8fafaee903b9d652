mkdir -p gpurun_out/r03o
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03o/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r03o/pytest.log | tail -2
timeout 600 python3 tools/oddtime.py > gpurun_out/r03o/oddtime.txt 2>&1
