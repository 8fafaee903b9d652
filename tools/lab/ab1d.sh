for rep in 1 2; do for v in a0 a1 cur; do
  if [ $v = cur ]; then unset PDWT_LIB_F32; else export PDWT_LIB_F32=$PWD/pypwt_amd/libalt_$v.so; fi
  echo -n "$v: "; python3 bench.py --config cfg3 --steps 50 --warmup 10 --no-cpu-baseline --no-extras | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f us'%(o['ms_per_step']*1e3), ' '.join('%s %.1f'%(k['kernel'],k['avg_us']) for k in o['kernels']))"
done; done
for w in db2 db4 db7 coif3; do for v in a0 a1 cur; do
  if [ $v = cur ]; then unset PDWT_LIB_F32; else export PDWT_LIB_F32=$PWD/pypwt_amd/libalt_$v.so; fi
  echo -n "$w $v: "; python3 - <<PY
import sys,time
sys.path.insert(0,".")
from pypwt_amd import BatchedWavelets
p=BatchedWavelets(1,1,1<<24,"$w",6,ndim=1); p.fill_hash(3)
def t(fn,n=200):
    for _ in range(20): fn()
    p.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    p.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("fwd %.1f fwd+inv %.1f"%(t(p.forward), t(lambda:(p.forward(),p.inverse()))))
PY
done; done
