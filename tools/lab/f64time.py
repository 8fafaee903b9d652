import time, numpy as np, sys
sys.path.insert(0, '.')
from pypwt_amd import Wavelets, Wavelets64
def run(cls, shape, wname, L, ndim=2, swt=0, dt=np.float32):
    rng = np.random.default_rng(1)
    x = (rng.random(shape) * 255).astype(dt)
    W = cls(x, wname, L, do_swt=swt, ndim=ndim)
    for _ in range(20): W.forward(); W.inverse()
    W.synchronize() if hasattr(W, 'synchronize') else None
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): W.forward(); W.inverse()
    W.synchronize() if hasattr(W, 'synchronize') else W.image
    t = (time.perf_counter() - t0) / n * 1e6
    byt = 16 * x.size * (2 if dt == np.float64 else 1)
    print(f"{cls.__name__:11s} {shape} {wname} L{L} swt={swt}: {t:8.1f} us/step  {byt/t/1e6:7.2f} TB/s algorithmic ({byt/t/8e6:.3f} of 8 TB/s)", flush=True)
run(Wavelets, (4096, 4096), 'db4', 4)
run(Wavelets64, (4096, 4096), 'db4', 4, dt=np.float64)
run(Wavelets, (1, 1 << 24), 'sym8', 6, ndim=1)
run(Wavelets64, (1, 1 << 24), 'sym8', 6, ndim=1, dt=np.float64)
run(Wavelets, (2048, 2048), 'haar', 5, swt=1)
run(Wavelets64, (2048, 2048), 'haar', 5, swt=1, dt=np.float64)
# filters of 10-20 taps in 2D: LDS tiles over real_t since round 3 (generic kernels before)
run(Wavelets, (4096, 4096), 'sym8', 4)
run(Wavelets64, (4096, 4096), 'sym8', 4, dt=np.float64)
run(Wavelets64, (4096, 4096), 'db6', 4, dt=np.float64)
run(Wavelets64, (4095, 4095), 'db4', 4, dt=np.float64)
