#!/bin/bash
set -u
OUT=gpurun_out/long7
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
tail -8 $OUT/pytest_gpu.txt
timeout 600 python3 tools/long_ab.py db20:4096x4096:3:1 db16:4096x4096:3:1 db15:4096x4096:3:1 db14:4096x4096:3:1 db13:4096x4096:3:1 db10:4096x4096:3:1 db9:4096x4096:3:1 db9:4096x4096:3:4 db10:4096x4096:3:4 db20:2048x2048:5:1 db15:2048x2048:5:1 db14:2048x2048:5:1 db20:4096x4096:3:16 > $OUT/long_ab.txt 2>&1
cat $OUT/long_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
o=json.loads(open("gpurun_out/long7/bench.json").read().strip().splitlines()[-1])
print("value",o["value"],"ms/step",o["ms_per_step"])
print("target",o.get("target"))
r=o["roofline"]; print({k:r[k] for k in ("kernel","avg_us","avg_us_basis","isolated_us","in_step_us","frac","event_overhead_us_per_launch")})
print(json.dumps(o["extra"].get("long_filters"),indent=1))
for k,v in o["extra"]["configs"].items(): print(k, {kk:v.get(kk) for kk in ("ms_per_step","frac_of_hbm_peak","dominant_kernel","dominant_kernel_us","dominant_kernel_us_basis","dominant_kernel_isolated_us","dominant_kernel_in_step_us","dominant_kernel_frac_of_hbm_peak")})
PY
