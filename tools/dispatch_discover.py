"""Prints the (launch name, kernel family) pairs every case of tests/dispatch_cases.py reaches at the default dispatch, and its
errors against the oracle (developer tool: the table of tests/test_gpu_dispatch.py was written from this output)."""
import sys
import time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle
from dispatch_cases import CASES, run_case

allp = {}
for case in CASES:
    t0 = time.time()
    try:
        pairs, cerr, rerr, L = run_case(case, oracle, np)
    except Exception as e:
        print(case, "FAILED", repr(e)[:300], flush=True)
        continue
    for pr in pairs:
        allp.setdefault((case[5],) + pr, []).append(case)
    print("%-70s L%d coeff rel err %.2e rec err %.2e  %.1fs  %s" % (case, L, cerr, rerr, time.time() - t0, sorted(pairs)), flush=True)
print()
for k in sorted(allp):
    print(k, len(allp[k]))
