import sys
sys.path.insert(0,'/root/repo')
from pypwt_amd import BatchedWavelets
for wname,L,shape in (("db20",5,(2048,2048)),("db10",3,(2048,2048)),("sym8",3,(2048,2048)),("db4",3,(2048,2048)),("db20",4,(1024,1024))):
    p=BatchedWavelets(1,shape[0],shape[1],wname,L,do_swt=1); p.fill_hash(5)
    for _ in range(3): p.forward(); p.inverse()
    p.synchronize()
    p.enable_kernel_timing(True); p.reset_kernel_times()
    for _ in range(20): p.forward(); p.inverse()
    p.synchronize()
    kt=p.kernel_times(); fam=p.kernel_families()
    n=len(kt)//20
    print(wname,L,shape)
    for i in range(n):
        ts=[kt[i+n*j][1] for j in range(20)]
        print("   %-22s %-8s %7.1f us"%(kt[i][0],fam[i],sum(ts)/len(ts)))
