#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PDWT_DWT_SPLIT_FWD=110 PDWT_DWT_SPLIT_INV=110 PDWT_NO_PYRAMID=1 PDWT_DSPLIT_CH=4
for nt in 64 256 512 1024; do
  export PDWT_DSPLIT_NT=$nt
  rm -rf gpurun_out/ab_tmp
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_tmp -- python3 tools/ktimes.py $1 $2 $3 $4 > /dev/null 2>&1
  echo "== $1 $2x$3 L$4 NT=$nt"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/ab_tmp/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "dwt_row" not in n: continue
    n=n.split("pdwt::")[1].split("(")[0]
    d[(n,int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:(kv[0][0].split("<")[0],-kv[0][1])):
    v.sort(); print("  %-44s grid=%8d med=%7.2f min=%7.2f"%(k[0],k[1],v[len(v)//2],v[0]))
PY
done
rm -rf gpurun_out/ab_tmp
