mkdir -p gpurun_out/r03t
PDWT_NO_LDS_TILES=1 timeout 600 python3 - > gpurun_out/r03t/f64_before.txt 2>&1 <<'PY'
import sys, numpy as np
sys.argv=["x"]; sys.path.insert(0,"."); sys.path.insert(0,"tools")
src=open("tools/f64time.py").read().split("run(Wavelets, (4096, 4096), 'db4', 4)")[0]
exec(src)
run(Wavelets64, (4096, 4096), 'sym8', 4, dt=np.float64)
run(Wavelets64, (4096, 4096), 'db6', 4, dt=np.float64)
run(Wavelets64, (4095, 4095), 'db4', 4, dt=np.float64)
run(Wavelets64, (2048, 2048), 'db10', 4, dt=np.float64)
PY
timeout 600 python3 - > gpurun_out/r03t/f64_after.txt 2>&1 <<'PY'
import sys, numpy as np
sys.argv=["x"]; sys.path.insert(0,"."); sys.path.insert(0,"tools")
src=open("tools/f64time.py").read().split("run(Wavelets, (4096, 4096), 'db4', 4)")[0]
exec(src)
run(Wavelets64, (4096, 4096), 'sym8', 4, dt=np.float64)
run(Wavelets64, (4096, 4096), 'db6', 4, dt=np.float64)
run(Wavelets64, (4095, 4095), 'db4', 4, dt=np.float64)
run(Wavelets64, (2048, 2048), 'db10', 4, dt=np.float64)
PY
