#!/usr/bin/env python3
"""What is dispatched when: the launch lists build_schedule (pypwt_amd/csrc/plan.cpp) produces for representative plans,
printed as a markdown table (DESIGN.md section 3 quotes it).   python3 tools/dispatch_table.py > profiles/r03_dispatch_table.md

A step is one launch: LEVEL = one level (2D: LDS tile / wave kernel by size, launch_dwt2.hip), PYR2 / PYR3 = two / three
levels per launch in LDS (small levels), STRIP2 = two levels per launch streaming down column strips (batches), REG1D = up
to three 1D levels per launch in registers, FUSED1D = 1D levels out of LDS, SWTF = two or three 2-tap SWT levels per launch
in registers, CHAIN = levels chained inside one launch (opt-in)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets  # noqa: E402

CASES = [
    ("cfg1", 1, 512, 512, "db2", 3, 0, 2), ("cfg2", 1, 4096, 4096, "db4", 4, 0, 2), ("cfg2, batch 2", 2, 4096, 4096, "db4", 4, 0, 2),
    ("cfg2, batch 16", 16, 4096, 4096, "db4", 4, 0, 2), ("cfg3", 1, 1, 1 << 24, "sym8", 6, 0, 1), ("cfg4", 1, 2048, 2048, "haar", 5, 1, 2),
    ("2048^2 db4 L4", 1, 2048, 2048, "db4", 4, 0, 2), ("1024^2 db4 L3", 1, 1024, 1024, "db4", 3, 0, 2), ("256^2 db4 L5", 1, 256, 256, "db4", 5, 0, 2),
    ("4096^2 sym8 L4", 1, 4096, 4096, "sym8", 4, 0, 2), ("4096^2 db20 L4", 1, 4096, 4096, "db20", 4, 0, 2), ("4095^2 db4 L4", 1, 4095, 4095, "db4", 4, 0, 2),
    ("2048^2 db2 SWT L3 (doc/denoising.rst)", 1, 2048, 2048, "db2", 3, 1, 2), ("4 x 2048^2 haar SWT L5", 4, 2048, 2048, "haar", 5, 1, 2),
    ("1D SWT 2^24 db4 L5", 1, 1, 1 << 24, "db4", 5, 1, 1), ("4096 rows x 4096 sym8 L6 (batched 1D)", 1, 4096, 4096, "sym8", 6, 0, 1),
    ("1D 2^24 db20 L6", 1, 1, 1 << 24, "db20", 6, 0, 1),
    ("2048^2 haar L11 (test/benchmark.py: maximum levels)", 1, 2048, 2048, "haar", 11, 0, 2), ("128^2 haar L7", 1, 128, 128, "haar", 7, 0, 2),
    ("2048^2 db2 L9", 1, 2048, 2048, "db2", 9, 0, 2), ("4096 x 64^2 db4 L3 (batch of tiny images)", 4096, 64, 64, "db4", 3, 0, 2),
    ("256 x 256^2 db4 L3", 256, 256, 256, "db4", 3, 0, 2), ("2 x 4096^2 haar L4", 2, 4096, 4096, "haar", 4, 0, 2),
]

print("| plan | forward launches | inverse launches |")
print("|---|---|---|")
for name, B, r, c, w, L, swt, ndim in CASES:
    p = BatchedWavelets(B, r, c, w, L, do_swt=swt, ndim=ndim)
    fwd, inv = [l.split(":", 1)[1].strip() for l in p.schedule().strip().splitlines()]
    print("| %s | %s | %s |" % (name, fwd, inv))
    p.cleanup()
