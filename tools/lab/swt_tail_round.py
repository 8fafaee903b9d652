"""The tiled SWT level kernels run two 128 x 16 tiles per CU at a time: 512 workgroups are one round, 520 are two.  Per-launch times
with the 64 x 8 tiles (four times the workgroups) forced for launches of up to PDWT_SWT_SMALL_TILE workgroups (lab library; one
process per setting, same box).   python3 tools/swt_tail_round.py > profiles/r05l_swt_tail_round.txt"""
import os
import subprocess
import sys


def child():
    sys.path.insert(0, '.')
    from pypwt_amd import _lib
    _lib.use_lab_kernels(True)
    from pypwt_amd import BatchedWavelets
    lib = _lib.load()
    lib.pdwt_set_tuning(b"swt_split_inv", 0)
    lib.pdwt_set_tuning(b"swt_split_fwd", 0)
    for w in ("db4", "db6", "sym8"):
        for s in ((1024, 1024), (1040, 1024), (1200, 1000), (1024, 1280), (1080, 1920), (1440, 1440)):
            bw = BatchedWavelets(1, s[0], s[1], w, 3, do_swt=1)
            bw.fill_hash(1)
            for _ in range(10): bw.forward(); bw.inverse()
            bw.synchronize(); bw.enable_kernel_timing(True); bw.reset_kernel_times()
            for _ in range(20): bw.forward(); bw.inverse()
            t = bw.kernel_times(cap=4096)
            per = len(t) // 20
            v = [sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)[10] * 1e3 for i in range(per)]
            print("%s %dx%d %s" % (w, s[0], s[1], " ".join("%.1f" % x for x in v)), flush=True)
            bw.cleanup()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
        sys.exit(0)
    res = {}
    for name, env in (("256", {}), ("700", {"PDWT_SWT_SMALL_TILE": "700"}), ("1100", {"PDWT_SWT_SMALL_TILE": "1100"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout
        for l in out.splitlines():
            f = l.split()
            if len(f) >= 8:
                res.setdefault((f[0], f[1]), {})[name] = [float(x) for x in f[2:]]
    print("# wavelet shape | per launch (3 forward, 3 inverse levels, us incl. ~2.5 us of events): 64 x 8 tiles below 256 workgroups (default) | below 700 | below 1100")
    for k, r in res.items():
        print("%-5s %-10s | %s | %s | %s" % (k[0], k[1], *[" ".join("%5.1f" % x for x in r.get(n, [])) for n in ("256", "700", "1100")]))
