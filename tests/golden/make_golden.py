#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ from PyWavelets.

PyWavelets is the reference's own oracle (reference test/test_wavelets.py:17-38,
test/testutils.py:5-10): every forward transform of the reference is pinned to
pywt.wavedec2 / wavedec (mode="periodization") and swt2 / swt, every inverse to
perfect reconstruction.  pywt is only importable in the BUILD container
(/opt/conda/bin/python3.9, PyWavelets 1.1.1); it does not travel to the GPU
box, so its outputs are committed here as data.

Run:   /opt/conda/bin/python3.9 tests/golden/make_golden.py
Writes (all under tests/golden/):
  filters.json        72 wavelet banks (dec_lo, dec_hi, rec_lo, rec_hi) as float64
  small_cases.npz     seeded inputs + every pywt subband for a 12-wavelet subset
  digests.json        per-band float64 digests for all 72 wavelets (small shapes)
                      and for BASELINE.json configs 2-4 at full size
  cfg1.npz            config 1 in full: 512x512 db2 L3, seeded input, 10 subbands
  threshold.npz       pywt.threshold soft/hard vectors

Only data is written: inputs, expected outputs, digests.  No reference source.
"""
import json
import os
import sys
import time
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import pywt  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PER = "periodization"

# Same list and order as the reference's table (pdwt/src/filters.cpp:5919-6002 and
# test/testutils.py:123-195): 71 names + haar.
WNAMES = (
    ["db%d" % i for i in range(2, 21)]
    + ["sym%d" % i for i in range(2, 21)]
    + ["coif%d" % i for i in range(1, 6)]
    + ["bior1.3", "bior1.5", "bior2.2", "bior2.4", "bior2.6", "bior2.8", "bior3.1",
       "bior3.3", "bior3.5", "bior3.7", "bior3.9", "bior4.4", "bior5.5", "bior6.8"]
    + ["rbio1.3", "rbio1.5", "rbio2.2", "rbio2.4", "rbio2.6", "rbio2.8", "rbio3.1",
       "rbio3.3", "rbio3.5", "rbio3.7", "rbio3.9", "rbio4.4", "rbio5.5", "rbio6.8"]
    + ["haar"]
)
assert len(WNAMES) == 72

SUBSET = ["haar", "db2", "db4", "db20", "sym8", "sym20", "coif1", "coif5",
          "bior1.3", "bior3.1", "bior6.8", "rbio3.1"]


def hash_input(shape, seed, scale=255.0):
    """Counter-based input generator shared by fixtures, oracle, HIP and bench.

    u = lowbias32(i ^ seed); x = (u >> 8) * 2^-24 * scale  (float32).
    Same arithmetic in oracle/pdwt_oracle.c (oracle_fill_hash) and in the HIP
    kernel fill_hash_kernel, so the 64 GiB of config 5 never cross PCIe.
    """
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i ^ np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    x = (h >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0) * scale
    return x.astype(np.float32).reshape(shape)


def max_level(n, hlen):
    """Reference clamp: floor(log2(N / (hlen-1)))  (pdwt/src/wt.cu:155-165)."""
    q = n // (hlen - 1)
    l = 0
    while q > 1:
        q >>= 1
        l += 1
    return l


def digest(a):
    a = np.asarray(a, dtype=np.float64)
    flat = a.ravel()
    return {
        "shape": list(a.shape),
        "sum": float(flat.sum()),
        "sumabs": float(np.abs(flat).sum()),
        "sumsq": float((flat * flat).sum()),
        "min": float(flat.min()),
        "max": float(flat.max()),
        "sample": [float(v) for v in flat[::4099][:64]],
    }


# ----------------------------------------------------------------------------
# pywt drivers, arranged in the reference's coefficient order
#   2D: [A_L, (H1,V1,D1), ..., (HL,VL,DL)]   level 1 = finest (pypwt.pyx:187-205)
#   1D: [A_L, D1, ..., DL]
# ----------------------------------------------------------------------------
def pywt_dwt2(x, wname, levels):
    c = pywt.wavedec2(x.astype(np.float64), wname, mode=PER, level=levels)
    out = [c[0]]
    for i in range(levels):  # test_wavelets.py:245-255: W.coeffs[i+1] <-> Wpy[levels-i]
        out += [c[levels - i][0], c[levels - i][1], c[levels - i][2]]
    return out


def pywt_dwt1(x, wname, levels):
    """(batched) 1D along the last axis (test_wavelets.py:372)."""
    c = pywt.wavedec(x.astype(np.float64), wname, mode=PER, level=levels, axis=-1)
    return [c[0]] + [c[levels - i] for i in range(levels)]


def pywt_swt2(x, wname, levels):
    # PyWavelets >= 1.0 returns the COARSEST level first; map to the
    # reference layout (level 1 = finest first).
    c = pywt.swt2(x.astype(np.float64), wname, level=levels)
    out = [c[0][0]]
    for i in range(levels):
        cA, (cH, cV, cD) = c[levels - 1 - i]
        out += [cH, cV, cD]
    return out


def pywt_swt1(x, wname, levels):
    c = pywt.swt(x.astype(np.float64), wname, level=levels, axis=-1)
    out = [c[0][0]]
    for i in range(levels):
        out.append(c[levels - 1 - i][1])
    return out


def main():
    t00 = time.time()
    # ------------------------------------------------------------------ filters
    filt = {}
    for w in WNAMES:
        W = pywt.Wavelet(w)
        filt[w] = {
            "hlen": W.dec_len,
            "dec_lo": [float(v) for v in W.dec_lo],
            "dec_hi": [float(v) for v in W.dec_hi],
            "rec_lo": [float(v) for v in W.rec_lo],
            "rec_hi": [float(v) for v in W.rec_hi],
        }
    with open(os.path.join(HERE, "filters.json"), "w") as f:
        json.dump({"pywt_version": pywt.__version__, "order": WNAMES, "filters": filt}, f, indent=0)

    # -------------------------------------------------------------- small cases
    small = {}
    meta = []
    seed = 1000

    def add_case(kind, wname, shape, levels, bands, x):
        nonlocal seed
        key = "c%03d" % len(meta)
        meta.append({"key": key, "kind": kind, "wname": wname, "shape": list(shape),
                     "levels": levels, "nbands": len(bands), "seed": seed})
        small[key + "_x"] = x
        for b, arr in enumerate(bands):
            small["%s_b%d" % (key, b)] = np.asarray(arr, dtype=np.float32)

    for w in SUBSET:
        hlen = pywt.Wavelet(w).dec_len
        # 2D DWT: even and odd shapes, level 1 and max-clamped level
        for shape in [(64, 64), (61, 59)]:
            lm = max(1, max_level(min(shape), hlen))
            for lv in sorted(set([1, lm])):
                seed += 1
                x = hash_input(shape, seed)
                add_case("dwt2", w, shape, lv, pywt_dwt2(x, w, lv), x)
        # 1D and batched 1D DWT
        for shape in [(1, 256), (1, 251), (8, 128)]:
            lm = max(1, max_level(shape[1], hlen))
            for lv in sorted(set([1, lm])):
                seed += 1
                x = hash_input(shape, seed)
                add_case("dwt1", w, shape, lv, pywt_dwt1(x, w, lv), x)
        # SWT (even sizes only: pywt cannot do odd, test_wavelets.py:193-204)
        lm = max(1, min(3, max_level(64, hlen)))
        for lv in sorted(set([1, lm])):
            seed += 1
            x = hash_input((64, 64), seed)
            add_case("swt2", w, (64, 64), lv, pywt_swt2(x, w, lv), x)
        for shape in [(1, 256), (8, 128)]:
            lm = max(1, min(4, max_level(shape[1], hlen)))
            for lv in sorted(set([1, lm])):
                seed += 1
                x = hash_input(shape, seed)
                add_case("swt1", w, shape, lv, pywt_swt1(x, w, lv), x)
    small["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **small)

    # ----------------------------------- ISWT as a linear operator:
    # arbitrary (non-transform) coefficients -> pywt.iswt2 / iswt
    isw = {}
    imeta = []
    for w in ["haar", "db2", "db4", "sym8", "bior2.2", "coif1"]:
        for lv in (1, 2):
            seed += 1
            shape = (32, 32)
            bands = [hash_input(shape, seed * 7 + b, 2.0) - 1.0 for b in range(3 * lv + 1)]
            # pywt input order: coarsest first; only the coarsest cA is used
            c = []
            for i in range(lv):
                lvl = lv - i  # level number of this entry
                H, V, D = bands[3 * (lvl - 1) + 1], bands[3 * (lvl - 1) + 2], bands[3 * (lvl - 1) + 3]
                cA = bands[0] if i == 0 else np.zeros(shape)
                c.append((cA.astype(np.float64), (H.astype(np.float64), V.astype(np.float64), D.astype(np.float64))))
            rec = pywt.iswt2(c, w)
            key = "i%02d" % len(imeta)
            imeta.append({"key": key, "kind": "iswt2", "wname": w, "shape": list(shape), "levels": lv,
                          "nbands": len(bands)})
            for b, arr in enumerate(bands):
                isw["%s_b%d" % (key, b)] = arr.astype(np.float32)
            isw[key + "_rec"] = rec.astype(np.float32)
            # 1D
            seed += 1
            shape1 = (1, 128)
            bands1 = [hash_input(shape1, seed * 11 + b, 2.0) - 1.0 for b in range(lv + 1)]
            c1 = []
            for i in range(lv):
                lvl = lv - i
                cA = bands1[0][0] if i == 0 else np.zeros(shape1[1])
                c1.append((cA.astype(np.float64), bands1[lvl][0].astype(np.float64)))
            rec1 = pywt.iswt(c1, w)
            key = "i%02d" % len(imeta)
            imeta.append({"key": key, "kind": "iswt1", "wname": w, "shape": list(shape1), "levels": lv,
                          "nbands": len(bands1)})
            for b, arr in enumerate(bands1):
                isw["%s_b%d" % (key, b)] = arr.astype(np.float32)
            isw[key + "_rec"] = rec1.astype(np.float32).reshape(shape1)
    isw["meta_json"] = np.frombuffer(json.dumps(imeta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "iswt_cases.npz"), **isw)

    # ------------------------------------------------ digests for all 72 names
    dig = {"all_wavelets": [], "configs": {}}
    for w in WNAMES:
        hlen = pywt.Wavelet(w).dec_len
        for kind, shape in [("dwt2", (64, 64)), ("dwt2", (61, 59)), ("dwt1", (1, 256)),
                            ("dwt1", (3, 251)), ("swt2", (32, 32)), ("swt1", (2, 128))]:
            n = min(shape) if kind.endswith("2") else shape[1]
            lm = max(1, max_level(n, hlen))
            if kind.startswith("swt"):
                lm = min(lm, 2)
            seed += 1
            x = hash_input(shape, seed)
            fn = {"dwt2": pywt_dwt2, "dwt1": pywt_dwt1, "swt2": pywt_swt2, "swt1": pywt_swt1}[kind]
            bands = fn(x, w, lm)
            dig["all_wavelets"].append({
                "kind": kind, "wname": w, "shape": list(shape), "levels": lm, "seed": seed,
                "bands": [{"sum": float(np.sum(b)), "sumabs": float(np.abs(b).sum()),
                           "sumsq": float((np.asarray(b) ** 2).sum()), "shape": list(np.shape(b))}
                          for b in bands]})

    # ------------------------------------------------------------ cfg1 in full
    x1 = hash_input((512, 512), 20241)
    b1 = pywt_dwt2(x1, "db2", 3)
    cfg1 = {"x": x1}
    for i, b in enumerate(b1):
        cfg1["b%d" % i] = np.asarray(b, dtype=np.float32)
    try:
        from scipy.misc import ascent
        asc = ascent().astype(np.uint8)
        cfg1["ascent_u8"] = asc
        ba = pywt_dwt2(asc.astype(np.float32), "db2", 3)
        for i, b in enumerate(ba):
            cfg1["ascent_b%d" % i] = np.asarray(b, dtype=np.float32)
    except Exception as e:  # pragma: no cover
        print("ascent unavailable:", e)
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), **cfg1)

    # ------------------------------------------------------- cfg2-4 as digests
    timing = {}
    # cfg2: 4096^2 db4 L4
    x2 = hash_input((4096, 4096), 20242)
    t0 = time.time(); b2 = pywt_dwt2(x2, "db4", 4); timing["cfg2_fwd_s"] = time.time() - t0
    dig["configs"]["cfg2"] = {"kind": "dwt2", "wname": "db4", "shape": [4096, 4096], "levels": 4,
                              "seed": 20242, "bands": [digest(b) for b in b2]}
    del b2
    # cfg3: 1D 2^24 sym8 L6
    x3 = hash_input((1, 1 << 24), 20243)
    t0 = time.time(); b3 = pywt_dwt1(x3, "sym8", 6); timing["cfg3_fwd_s"] = time.time() - t0
    dig["configs"]["cfg3"] = {"kind": "dwt1", "wname": "sym8", "shape": [1, 1 << 24], "levels": 6,
                              "seed": 20243, "bands": [digest(b) for b in b3]}
    del b3
    # cfg4: SWT2 2048^2 haar L5 + soft threshold beta = 0.1 * range
    x4 = hash_input((2048, 2048), 20244)
    t0 = time.time(); b4 = pywt_swt2(x4, "haar", 5); timing["cfg4_fwd_s"] = time.time() - t0
    beta = 25.5
    dig["configs"]["cfg4"] = {"kind": "swt2", "wname": "haar", "shape": [2048, 2048], "levels": 5,
                              "seed": 20244, "beta": beta,
                              "bands": [digest(b) for b in b4],
                              "bands_soft": [digest(b if i == 0 else pywt.threshold(b, beta, "soft"))
                                             for i, b in enumerate(b4)]}
    # reconstruction after threshold (pywt.iswt2 is the linear operator PDWT's ISWT equals)
    c = []
    for i in range(5):
        lvl = 5 - i
        H, V, D = [pywt.threshold(b4[3 * (lvl - 1) + 1 + k], beta, "soft") for k in range(3)]
        cA = b4[0] if i == 0 else np.zeros_like(b4[0])
        c.append((cA, (H, V, D)))
    t0 = time.time(); rec4 = pywt.iswt2(c, "haar"); timing["cfg4_inv_s"] = time.time() - t0
    dig["configs"]["cfg4"]["rec_soft"] = digest(rec4)
    del b4, c, rec4
    dig["pywt_timing_build_container"] = timing
    dig["pywt_version"] = pywt.__version__
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(dig, f)

    # ------------------------------------------------------------- thresholds
    xt = hash_input((4096,), 777, 40.0) - 20.0
    th = {"x": xt}
    for k, beta in enumerate([0.0, 0.1, 10.0]):
        th["beta%d" % k] = np.float32(beta)
        th["soft%d" % k] = pywt.threshold(xt.astype(np.float64), beta, "soft").astype(np.float32)
        th["hard%d" % k] = pywt.threshold(xt.astype(np.float64), beta, "hard").astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "threshold.npz"), **th)
    print("golden vectors written in %.1f s" % (time.time() - t00))


if __name__ == "__main__":
    sys.exit(main())
