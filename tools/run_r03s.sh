mkdir -p gpurun_out/r03s
timeout 1500 python3 -m pytest tests -q -m gpu -k "fp64 or f64 or Wavelets64 or 64" > gpurun_out/r03s/pytest_f64.log 2>&1; grep -E "passed|failed" gpurun_out/r03s/pytest_f64.log | tail -2
timeout 600 python3 tools/f64time.py > gpurun_out/r03s/f64time.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu > gpurun_out/r03s/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r03s/pytest.log | tail -2
