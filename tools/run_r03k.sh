mkdir -p gpurun_out/r03k
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ops.py -x -q -k "swt or 72 or fuzz or fp64" > gpurun_out/r03k/pytest_swt.log 2>&1; tail -3 gpurun_out/r03k/pytest_swt.log
timeout 900 python3 tools/refbench.py > gpurun_out/r03k/refbench.txt 2> gpurun_out/r03k/refbench.err
PDWT_NO_SWT1_VEC=1 timeout 300 python3 - > gpurun_out/r03k/swt1_old.txt 2>&1 <<'PY'
import sys; sys.argv=["x"]
sys.path.insert(0,"tools"); sys.path.insert(0,".")
import refbench
for w in ("haar","db4","sym8"): refbench.case("swt1", w, (1, 1<<24), levels=5, inverse_too=True)
refbench.case("swt1", "db4", (4096, 4096), levels=5, inverse_too=True)
PY
echo done
