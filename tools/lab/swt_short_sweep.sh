#!/bin/bash
# 2D / 1D SWT of SHORT filters with the two-launch (register-blocked row + column) levels forced off / on
for sz in "2048 2048 3" "1024 1024 3" "512 512 3"; do python3 tools/swtsweep.py $sz haar,db2,db3,db4,db5; done
echo "# 1D SWT 2^24 L4, inverse threshold PDWT_SWT1_SPLIT_INV = 10 | 4"
for w in db2 db3 db4; do for t in 10 4; do echo -n "$w inv>=$t: "; PDWT_SWT1_SPLIT_INV=$t python3 - <<PY
import sys,time
sys.path.insert(0,".")
from pypwt_amd import BatchedWavelets
for shape in ((1,1<<24),(4096,4096),(1,1<<20)):
    p=BatchedWavelets(1,shape[0],shape[1],"$w",4,do_swt=1,ndim=1); p.fill_hash(3)
    def t(fn,n=30):
        for _ in range(3): fn()
        p.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        p.synchronize(); return (time.perf_counter()-t0)/n*1e6
    tf=t(p.forward); tfi=t(lambda:(p.forward(),p.inverse()))
    print("%s fwd %.1f inv %.1f |"%(shape,tf,tfi-tf),end=" ")
print()
PY
done; done
