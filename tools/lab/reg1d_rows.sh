#!/bin/bash
# where the 1D register kernels (three levels per launch, overlapping 1024-sample blocks) beat the LDS pyramids (all levels in
# one launch): row length x total size, forward+inverse pipelined, reg | fused (PDWT_REG1D=3 | 0)
for w in ${WAVES:-haar db4 sym8}; do
python3 - <<PY
import sys,time,os,subprocess,json
sys.path.insert(0,".")
code='''
import sys,time
sys.path.insert(0,".")
from pypwt_amd import BatchedWavelets
out=[]
for rows,n in SHAPES:
    p=BatchedWavelets(1,rows,n,"$w",LEV,ndim=1); p.fill_hash(3)
    def t(fn,k=100):
        for _ in range(10): fn()
        p.synchronize(); t0=time.perf_counter()
        for _ in range(k): fn()
        p.synchronize(); return (time.perf_counter()-t0)/k*1e6
    out.append(t(lambda:(p.forward(),p.inverse())))
    p.cleanup()
print(out)
'''
shapes=[(4096,4096),(1024,16384),(256,65536),(64,1<<18),(16,1<<20),(4,1<<22),(1,1<<24),(1024,4096),(64,65536),(1,1<<22),(256,4096),(16,65536),(1,1<<20),(64,4096),(1,1<<18)]
res={}
for r in ("3","0"):
    env=dict(os.environ,PDWT_REG1D=r)
    o=subprocess.run([sys.executable,"-c",code.replace("SHAPES",repr(shapes)).replace("LEV","5")],env=env,capture_output=True,text=True).stdout.strip().splitlines()[-1]
    res[r]=eval(o)
print("$w  (rows x samples: reg | fused us, fwd+inv L5)")
for i,s in enumerate(shapes):
    a,b=res["3"][i],res["0"][i]
    print("   %5d x %-9d total 2^%2d   %7.1f | %7.1f   %s"%(s[0],s[1],(s[0]*s[1]).bit_length()-1,a,b,"reg" if a<b else "FUSED"))
PY
done
