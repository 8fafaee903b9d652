"""Extra seeds for the GPU fuzz tests (developer tool): python tools/morefuzz.py first count"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_fuzz as t
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    for fn in (t.test_fuzz_against_oracle, t.test_fuzz_batched_plans, t.test_fuzz_register_kernels, t.test_fuzz_three_level_pyramid):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", fn.__name__, seed, str(e)[:300], flush=True)
print("done, failures:", bad)
