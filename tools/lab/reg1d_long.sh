#!/bin/bash
# 1D DWT of filters beyond 20 taps: register kernels (dwt1_reg_kernels.hpp, extended to 40 taps in round 4) against the LDS
# pyramids (PDWT_REG1D=0)
for w in ${WAVES:-db10 db11 db13 db16 db20}; do for r in 3 0; do
echo -n "$w reg1d=$r: "; PDWT_REG1D=$r python3 - <<PY
import sys,time
sys.path.insert(0,".")
from pypwt_amd import BatchedWavelets
for shape,L in (((1,1<<24),6),((4096,4096),5),((1,1<<20),5),((512,2048),4)):
    p=BatchedWavelets(1,shape[0],shape[1],"$w",L,ndim=1); p.fill_hash(3)
    def t(fn,n=100):
        for _ in range(10): fn()
        p.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        p.synchronize(); return (time.perf_counter()-t0)/n*1e6
    tf=t(p.forward); tfi=t(lambda:(p.forward(),p.inverse()))
    print("%s fwd %.1f f+i %.1f |"%("x".join(map(str,shape)),tf,tfi),end=" ")
print(p.schedule().split("\n")[0])
PY
done; done
