#!/usr/bin/env python3
"""Kernels of a built library that spill to scratch (private_segment_fixed_size > 0), from the code objects embedded in the .so:
a spilling kernel is almost always a performance cliff (round 4: the fp64 SWT inverse of 12-24 taps, nine times the fp32 time).

    python3 tools/spillscan.py pypwt_amd/libpypwt_amd.so pypwt_amd/libpypwt_amd_f64.so
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def scan(path):
    data = open(path, "rb").read()
    out = []
    with tempfile.TemporaryDirectory() as td:
        n = 0
        for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
            i = m.start()
            if struct.unpack_from("<H", data, i + 18)[0] != 224:  # EM_AMDGPU
                continue
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            f = os.path.join(td, "co%d.elf" % n)
            open(f, "wb").write(data[i:i + shoff + shentsize * shnum])
            n += 1
            notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith(".name:"):
                    name = line.split(":", 1)[1].strip()
                elif line.startswith(".private_segment_fixed_size:"):
                    out.append((int(line.split(":")[1]), name))
    return out


for lib in sys.argv[1:]:
    ks = scan(lib)
    bad = sorted([k for k in ks if k[0] > 0], reverse=True)
    print("%s: %d kernels, %d with scratch" % (lib, len(ks), len(bad)))
    for size, name in bad:
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("  %6d B  %s" % (size, demangled[:150]))
