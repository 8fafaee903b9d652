mkdir -p gpurun_out/r03l
for ty in 16 32 64; do echo "== PDWT_SWT_TY=$ty"; PDWT_SWT_TY=$ty timeout 600 python3 - <<'PY'
import sys; sys.argv=["x"]
sys.path.insert(0,"tools"); sys.path.insert(0,".")
import refbench
for w in ("db5","db6","sym8","db12","db20"): refbench.case("swt2", w, (2048, 2048), levels=3, inverse_too=True)
refbench.case("swt2", "db20", (1024, 1024), levels=4, inverse_too=True)
PY
done > gpurun_out/r03l/swt_ty.txt 2>&1
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "swt" > gpurun_out/r03l/pytest.log 2>&1; tail -2 gpurun_out/r03l/pytest.log
