#!/bin/bash
# A/B builds of the fp64 library: tools/f64_variant.sh NAME -DPDWT_STREAM_R_A1=8 ...  ->  pypwt_amd/variants/libpypwt_amd_f64_NAME.so
# (only launch_swt_split.hip is recompiled; the other objects are the product build's).  Use with PDWT_LIB_F64=<that file>.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
NAME="$1"; shift
python3 -m pypwt_amd.build > /dev/null
mkdir -p pypwt_amd/variants build/obj_f64_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function --offload-arch=gfx950 -DPDWT_DOUBLE "$@" \
    -c pypwt_amd/csrc/launch_swt_split.hip -o build/obj_f64_$NAME/launch_swt_split.o
OBJS=$(python3 -c "
from pypwt_amd import build as b
import os
objdir, lib, extra, skip = b.VARIANTS['f64']
print(' '.join(os.path.join(objdir, s.rsplit('.', 1)[0] + '.o') for s in b.SOURCES if s not in skip and s != 'launch_swt_split.hip'))")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o pypwt_amd/variants/libpypwt_amd_f64_$NAME.so $OBJS build/obj_f64_$NAME/launch_swt_split.o 2>&1 | tail -3
echo pypwt_amd/variants/libpypwt_amd_f64_$NAME.so
