for s in 256 512 768 1024 1536 2048 4096; do echo "== slots $s"; PDWT_STRIP_SLOTS=$s timeout 200 python3 - <<'PY' 2>/dev/null | grep -v "^Warn\|^Forc"
import sys, time
sys.path.insert(0,'/root/repo')
from pypwt_amd import BatchedWavelets, _lib
_lib.use_lab_kernels(True)
def timed(fn, sync, n):
    for _ in range(3): fn()
    sync(); best=1e9
    for _ in range(3):
        t0=time.perf_counter()
        for _ in range(n): fn()
        sync(); best=min(best,(time.perf_counter()-t0)/n)
    return best*1e6
out=[]
for wname,shape,L in (("db4",(1080,1920),3),("sym8",(1080,1920),3),("db4",(1024,1024),3),("sym8",(2048,2048),3),("db4",(600,800),3),("db10",(1080,1920),3)):
    p=BatchedWavelets(1,shape[0],shape[1],wname,L,do_swt=1); p.fill_hash(5)
    def step():
        p.forward(); p.inverse()
    out.append("%s %dx%d fwd %.1f fwd+inv %.1f"%(wname,shape[0],shape[1],timed(p.forward,p.synchronize,100),timed(step,p.synchronize,100)))
print(" | ".join(out))
PY
done
