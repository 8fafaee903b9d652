"""Is a fused SWT forward launched image by image faster than the per-level launches a batch beyond the Infinity Cache gets?
(developer tool: N one-image plans on ONE stream against one N-image plan)"""
import sys
import time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets, _lib

lib = _lib.load()


def t(fn, sync, n=60):
    for _ in range(10):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


for wname, L in (("haar", 3), ("db2", 3), ("haar", 5)):
    for N in (2, 4):
        big = BatchedWavelets(N, 2048, 2048, wname, L, do_swt=1)
        big.fill_hash(1)
        tb_f = t(big.forward, big.synchronize)
        tb = t(lambda: (big.forward(), big.inverse()), big.synchronize)
        forced = None
        prev = lib.pdwt_set_tuning(b"swt_fused", 2)
        forced = BatchedWavelets(N, 2048, 2048, wname, L, do_swt=1)
        lib.pdwt_set_tuning(b"swt_fused", prev)
        forced.fill_hash(1)
        tf_f = t(forced.forward, forced.synchronize)
        stream = big._lib.pdwt_get_stream(big._h)
        singles = [BatchedWavelets(1, 2048, 2048, wname, L, do_swt=1, stream=stream) for _ in range(N)]
        for s_ in singles:
            s_.fill_hash(1)

        def fwd_all():
            for s_ in singles:
                s_.forward()

        def both_all():
            for s_ in singles:
                s_.forward()
            for s_ in singles:
                s_.inverse()
        ts_f = t(fwd_all, big.synchronize)
        ts = t(both_all, big.synchronize)
        print("%-5s L%d N=%d  one plan: fwd %7.1f fwd+inv %7.1f | fused forced: fwd %7.1f | %d one-image plans on one stream: fwd %7.1f fwd+inv %7.1f   [%s]"
              % (wname, L, N, tb_f, tb, tf_f, N, ts_f, ts, big.schedule().replace("\n", " | ")), flush=True)
        for p in [big, forced] + singles:
            p.cleanup()
