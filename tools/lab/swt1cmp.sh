for w in db5 sym8 db10 db20; do for e in 0 10; do echo "== $w PDWT_SWT1_SPLIT=$e"; PDWT_SWT1_SPLIT=$e python - <<PY
import sys, time, numpy as np
sys.path.insert(0, ".")
from pypwt_amd import Wavelets
for shape in ((1, 1 << 24), (4096, 4096)):
    x = (np.random.RandomState(1).rand(*shape) * 255).astype(np.float32)
    W = Wavelets(x, "$w", 5, do_swt=1, ndim=1)
    for _ in range(5): W.forward()
    W.synchronize(); t0 = time.perf_counter()
    for _ in range(30): W.forward()
    W.synchronize(); tf = (time.perf_counter() - t0) / 30 * 1e6
    W.synchronize(); t0 = time.perf_counter()
    for _ in range(30): W.forward(); W.inverse()
    W.synchronize(); tfi = (time.perf_counter() - t0) / 30 * 1e6
    print(shape, "L=%d" % W.levels, "fwd %.1f us  fwd+inv %.1f us" % (tf, tfi))
PY
done; done
