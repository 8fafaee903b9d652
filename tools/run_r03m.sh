mkdir -p gpurun_out/r03m
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "swt or 72 or fuzz" > gpurun_out/r03m/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r03m/pytest.log | tail -2
timeout 600 python3 - > gpurun_out/r03m/swt_long.txt 2>&1 <<'PY'
import sys; sys.argv=["x"]
sys.path.insert(0,"tools"); sys.path.insert(0,".")
import refbench
for w in ("sym8","db12","db14","db20"): refbench.case("swt2", w, (2048, 2048), levels=3, inverse_too=True)
refbench.case("swt2", "db20", (2048, 2048), levels=999)
refbench.case("swt2", "db20", (1024, 1024), levels=999)
refbench.case("dwt2", "db20", (2048, 2048), levels=999, inverse_too=True)
PY
