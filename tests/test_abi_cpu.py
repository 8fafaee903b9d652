"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/pypwt_amd.h declares; the built-in filter table equals the pywt vectors; argument
errors are reported without a GPU; nothing in the product imports the oracle.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pypwt_amd.build import build_library
    build_library(verbose=False)
    from pypwt_amd import _lib
    return _lib.load()


def header_functions(which=("pypwt_amd.h", "pypwt_amd_bench.h")):
    """Every function the headers declare: the contract (pypwt_amd.h) and the measurement / test hooks (pypwt_amd_bench.h)."""
    names = set()
    for h in which:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(pdwt_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


BENCH_ONLY = {"pdwt_fill_image_hash", "pdwt_enable_kernel_timing", "pdwt_kernel_times", "pdwt_reset_kernel_times", "pdwt_time_level",
              "pdwt_time_copy", "pdwt_copy_capacity", "pdwt_schedule_string", "pdwt_set_tuning", "pdwt_kernel_families"}


def test_the_contract_header_holds_no_measurement_hook():
    """an integrator reads pypwt_amd.h and finds the reference's members (plus the batch / stream / device
    plumbing); timing, micro-benchmarks, the synthetic input and the dispatch knobs live in pypwt_amd_bench.h."""
    contract, bench = set(header_functions(("pypwt_amd.h",))), set(header_functions(("pypwt_amd_bench.h",)))
    assert not (contract & BENCH_ONLY), contract & BENCH_ONLY
    assert bench == BENCH_ONLY, bench ^ BENCH_ONLY
    assert '#include "pypwt_amd.h"' in open(os.path.join(ROOT, "include", "pypwt_amd_bench.h")).read()
    src = open(os.path.join(ROOT, "tests", "c_abi", "roundtrip.c")).read()
    assert "pypwt_amd_bench.h" not in src  # the plain-C client of the contract needs none of it


def test_library_exports_every_header_symbol(lib):
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libpypwt_amd.so does not export %s" % n


def test_fp64_library_exports_the_same_symbols_with_double_taps():
    """libpypwt_amd_f64.so = the same sources with -DPDWT_DOUBLE (pdwt_real = double): every header
    symbol is there and the built-in taps keep their float64 precision."""
    from pypwt_amd.build import build_library
    build_library(verbose=False, variant="f64")
    from pypwt_amd import _lib
    from oracle import oracle
    lib64 = _lib.load("f64")
    for n in header_functions():
        assert hasattr(lib64, n), "libpypwt_amd_f64.so does not export %s" % n
    t = oracle.filter_table()
    buf = (C.c_double * 160)()
    for w in ("db4", "sym8", "bior6.8", "coif5"):
        hlen = lib64.pdwt_wavelet_filters(w.encode(), buf, 160)
        e = t["filters"][w]
        got = np.frombuffer(buf, dtype=np.float64, count=4 * hlen).reshape(4, hlen)
        want = np.array([e["dec_lo"], e["dec_hi"], e["rec_lo"], e["rec_hi"]], dtype=np.float64)
        assert np.array_equal(got, want), w


def test_python_binding_covers_the_header(lib):
    from pypwt_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_every_entry_point_cites_the_reference():
    """include/*.h must cite the reference interface each entry point replaces (file:line)."""
    txt = open(os.path.join(ROOT, "include", "pypwt_amd.h")).read()
    assert txt.count("wt.cu:") >= 15 and "src/pypwt.pyx:8-61" in txt and "wt.h:20-76" in txt


def test_filter_table_equals_pywt_vectors(lib):
    from oracle import oracle
    t = oracle.filter_table()
    assert lib.pdwt_wavelet_count() == 72
    names = [lib.pdwt_wavelet_name(i).decode() for i in range(72)]
    assert names == t["order"]  # the reference's table order (filters.cpp:5919-6002)
    buf = (C.c_float * 160)()
    for w in names:
        hlen = lib.pdwt_wavelet_filters(w.encode(), buf, 160)
        e = t["filters"][w]
        assert hlen == e["hlen"]
        got = np.frombuffer(buf, dtype=np.float32, count=4 * hlen).reshape(4, hlen)
        want = np.array([e["dec_lo"], e["dec_hi"], e["rec_lo"], e["rec_hi"]], dtype=np.float32)
        assert np.array_equal(got, want), w
    # Haar aliases (separable.cu:24-28) and case-insensitivity (strcasecmp, separable.cu:33)
    for alias in (b"db1", b"bior1.1", b"rbior1.1", b"HAAR", b"Db4"):
        assert lib.pdwt_wavelet_filters(alias, None, 0) in (2, 8)
    assert lib.pdwt_wavelet_filters(b"nope", None, 0) == -2


def test_argument_errors_without_gpu(lib):
    from pypwt_amd import _lib
    h = _lib.handle_t()
    img = np.zeros((8, 8), dtype=np.float32)
    p = img.ctypes.data_as(_lib.f32p)
    assert lib.pdwt_create(p, 8, 8, b"nope", 1, 1, 1, 0, 0, 2, C.byref(h)) == _lib.ERR_WAVELET
    assert b"unknown wavelet" in lib.pdwt_last_error()
    assert lib.pdwt_create(p, 0, 8, b"db2", 1, 1, 1, 0, 0, 2, C.byref(h)) == _lib.ERR_ARG
    assert lib.pdwt_forward(None) == _lib.ERR_ARG
    assert lib.pdwt_destroy(None) == 0


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """On a machine without a HIP device the product must FAIL, never compute on the CPU."""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from pypwt_amd import Wavelets\n"
            "try:\n"
            "    Wavelets(np.zeros((16,16),np.float32),'db2',1); print('CREATED')\n"
            "except Exception as e: print('RAISED', type(e).__name__)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "RAISED PdwtError" in out


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/.  The product may
    MENTION the oracle in comments (which file restates a formula); it must not import, load or link it."""
    import subprocess
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pypwt_amd")):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".inc")):
                continue
            for line in open(os.path.join(dirpath, f), errors="replace"):
                code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                assert "import oracle" not in code and "from oracle" not in code, (f, line)
                assert "libpdwt_oracle" not in code and "pdwt_oracle.c" not in code, (f, line)
    # neither shared library depends on, or carries symbols of, the oracle
    for so in ("libpypwt_amd.so", "libpypwt_amd_f64.so"):
        path = os.path.join(ROOT, "pypwt_amd", so)
        if os.path.exists(path):
            dyn = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
            assert "oracle_" not in dyn, so
    # bench.py: the oracle appears only inside the cpu_baseline leg
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src.split("def cpu_baseline")[1].split("\ndef ")[0]
    assert "oracle" in body
    rest = src.replace(body, "")
    assert "import oracle" not in rest and "from oracle" not in rest and "oracle." not in rest


def test_sources_are_gfx950_only():
    """No CUDA shims, no dual CUDA/HIP paths, no Triton (north star)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pypwt_amd")):
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip", ".py")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                for bad in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "import triton", "hipify"):
                    assert bad not in src, (f, bad)


def test_plain_c_program_links_against_the_abi(lib, tmp_path):
    """The boundary is a C ABI: a C99 translation unit including only include/pypwt_amd.h must compile
    with gcc and link against libpypwt_amd.so (it runs on the GPU box in tests/test_gpu_ops.py)."""
    import subprocess
    exe = str(tmp_path / "roundtrip")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "roundtrip.c"), "-L", os.path.join(ROOT, "pypwt_amd"),
                           "-lpypwt_amd", "-Wl,-rpath," + os.path.join(ROOT, "pypwt_amd"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode in (0, 77), r.stdout  # 77: no HIP device here; unknown-wavelet check already ran


def test_tiled_wavelets_needs_a_gpu_and_says_so():
    """pypwt_amd/tiled.py (one image over several GPUs) has no CPU path either."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pypwt_amd.tiled import TiledWavelets
    with pytest.raises(RuntimeError, match="HIP device|import torch before"):
        TiledWavelets(np.zeros((64, 64), dtype=np.float32), "db2", 2)


def test_python_class_has_the_reference_api_surface():
    """Methods and properties of the reference's `cdef class Wavelets` (src/pypwt.pyx:93-615)
    exist on the drop-in class with the same constructor parameters."""
    import inspect
    from pypwt_amd import Wavelets, Wavelets64
    methods = ["info", "coeff_only", "set_image", "forward", "inverse", "soft_threshold", "hard_threshold", "shrink",
               "norm1", "norm2sq", "add_wavelet", "set_coeff", "set_wavelets_filters", "image_int_ptr", "coeff_int_ptr",
               "div2", "_checkarray", "_compute_sizes"]
    for cls in (Wavelets, Wavelets64):
        for m in methods:
            assert callable(getattr(cls, m)), m
        for prop in ("coeffs", "image"):
            assert isinstance(getattr(cls, prop), property), prop
        params = list(inspect.signature(cls.__init__).parameters)
        assert params[:8] == ["self", "img", "wname", "levels", "do_separable", "do_cycle_spinning", "do_swt", "ndim"]
    sig = inspect.signature(Wavelets.soft_threshold).parameters
    assert list(sig)[1:] == ["beta", "do_threshold_appcoeffs", "normalize"]
    assert sig["do_threshold_appcoeffs"].default == 0 and sig["normalize"].default == 0   # pypwt.pyx:362
    assert inspect.signature(Wavelets.shrink).parameters["do_threshold_appcoeffs"].default == 1  # pypwt.pyx:406


def test_tuning_keys_documented_in_the_header_exist_and_round_trip(lib):
    """pdwt_set_tuning needs no device: every key the header documents is accepted, returns the previous value and can be
    restored; an unknown key is an argument error."""
    txt = open(os.path.join(ROOT, "include", "pypwt_amd_bench.h")).read()
    doc = txt[txt.index("process-wide dispatch knobs"):txt.index("int pdwt_set_tuning")]
    keys = sorted(set(re.findall(r'"([a-z0-9_]+)"', doc)))
    assert {"wave_min_log2", "lds_max_log2", "reg1d", "swt_fused", "swt_split_fwd", "swt_split_inv", "chain", "wave2"} <= set(keys)
    for k in keys:
        prev = lib.pdwt_set_tuning(k.encode(), 1)
        assert prev >= 0, (k, prev)
        back = lib.pdwt_set_tuning(k.encode(), prev)
        # (the experiment kernels live in libpypwt_amd_lab.so only: their knobs are accepted by the product and do nothing)
        assert back == 1 or k == "chain_timeout", (k, back)
    assert lib.pdwt_set_tuning(b"no_such_knob", 1) < 0
    # the split-SWT thresholds: taps, 0 = never, 100 + n = n taps at every size
    p = lib.pdwt_set_tuning(b"swt_split_inv", 110)
    assert lib.pdwt_set_tuning(b"swt_split_inv", p) == 110


def test_partition_of_a_sharded_batch():
    """pypwt_amd.sharded: contiguous blocks, remainder to the first owners, every image owned exactly once (CPU only)."""
    from pypwt_amd.sharded import owner_of, partition_images
    assert partition_images(1024, 8) == [(128 * r, 128 * (r + 1)) for r in range(8)]  # BASELINE config 5
    assert partition_images(5, 2) == [(0, 3), (3, 5)]
    assert partition_images(2, 3) == [(0, 1), (1, 2), (2, 2)]
    for total, parts in ((1, 1), (7, 3), (130, 8), (3, 8)):
        blocks = partition_images(total, parts)
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(hi - lo for lo, hi in blocks) - min(hi - lo for lo, hi in blocks) <= 1
        for b in range(total):
            i, k = owner_of(blocks, b)
            assert blocks[i][0] + k == b
    import pytest
    with pytest.raises(IndexError):
        owner_of(partition_images(4, 2), 4)
    with pytest.raises(ValueError):
        partition_images(4, 0)


def test_sharded_batch_without_a_gpu_fails_loudly():
    import pytest
    from pypwt_amd import ShardedBatch, _lib
    if _lib.load().pdwt_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError):
        ShardedBatch(4, 64, 64, "db2", 2)


def test_no_kernel_of_the_product_libraries_uses_scratch():
    """A kernel that spills to scratch is a performance cliff (round 4: the fp64 SWT inverse of 12-24 taps ran nine times slower than
    the fp32 one).  tools/spillscan.py reads private_segment_fixed_size of every kernel out of the built libraries."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = [os.path.join(here, "pypwt_amd", n) for n in ("libpypwt_amd.so", "libpypwt_amd_f64.so")]
    if not all(os.path.exists(p) for p in libs) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("libraries not built / llvm-readelf not available")
    out = subprocess.run([sys.executable, os.path.join(here, "tools", "spillscan.py")] + libs, capture_output=True, text=True, timeout=600).stdout
    lines = [l for l in out.splitlines() if l.startswith(os.path.join(here, "pypwt_amd"))]
    assert len(lines) == 2 and all(l.endswith(" 0 with scratch") for l in lines), out


def test_product_library_reads_only_the_documented_environment():
    """53 getenv knobs lived in the shipped library.  Now the product sources read at most 12
    variables with getenv -- exactly those INTEGRATION.md section 5 lists first -- and every A/B knob goes through lab_env(), which
    is getenv in libpypwt_amd_lab.so (-DPDWT_LAB_KERNELS) and a constant "unset" in the product libraries."""
    import glob
    names = set()
    for f in glob.glob(os.path.join(ROOT, "pypwt_amd", "csrc", "*")):
        src = open(f).read()
        names |= set(re.findall(r'(?<![a-z_])getenv\("(PDWT_[A-Z0-9_]+)"\)', src))
        for line in src.splitlines():  # no getenv on a run-time name outside lab_env itself
            if re.search(r'(?<![a-z_])getenv\((?!")', line):
                assert "lab_env" in line and f.endswith("launch_util.hpp"), (f, line)
    assert 0 < len(names) <= 12, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 5. Environment variables"):]
    product_table = sec[:sec.index("**Everything below exists only")]
    documented = set(re.findall(r"`(PDWT_[A-Z0-9_]+)", product_table))
    assert names == documented, (names ^ documented)
    util = open(os.path.join(ROOT, "pypwt_amd", "csrc", "launch_util.hpp")).read()
    assert "#ifdef PDWT_LAB_KERNELS" in util and "static inline const char* lab_env(const char*) { return nullptr; }" in util


def test_unique_id_rendezvous_ignores_strangers_and_serves_every_rank_once():
    """pypwt_amd.comm._share_id (no GPU, no RCCL: just the TCP hand-off of the 128 id bytes).  A stray connection that does not
    say the hello line -- a port scanner, a rank of another job with another nonce -- gets nothing and does not use up a rank's
    turn (round 4 advice: the accept loop counted connections, so a stranger made a legitimate rank wait for the timeout)."""
    import socket
    import threading
    import time
    from pypwt_amd import comm
    os.environ["PDWT_COMM_NONCE"] = "job-a"
    port = 29777
    uid = bytes(range(128))
    got = {}

    def server():
        got[0] = comm._share_id(0, 3, uid, "127.0.0.1", port, timeout=30.0)

    t = threading.Thread(target=server)
    t.start()
    time.sleep(0.3)
    # strangers: silence, garbage, the right words with the wrong nonce
    for payload in (b"", b"GET / HTTP/1.0\r\n\r\n", comm.HELLO + b"1 job-b\n", comm.HELLO + b"7 job-a\n"):
        with socket.create_connection(("127.0.0.1", port), timeout=5.0) as s:
            if payload:
                s.sendall(payload)
            s.settimeout(1.0)
            try:
                assert s.recv(128) == b""  # nothing for a stranger: end of stream, a reset, or silence
            except (socket.timeout, ConnectionResetError):
                pass
    for r in (2, 1):
        got[r] = comm._share_id(r, 3, None, "127.0.0.1", port, timeout=30.0)
    t.join(30.0)
    assert not t.is_alive()
    assert got[0] == got[1] == got[2] == uid
    os.environ.pop("PDWT_COMM_NONCE", None)


def _host_ring_rank(rank, size, port, q):
    os.environ["PDWT_COMM_NONCE"] = "ring-test"
    from pypwt_amd.comm import HostRing
    ring = HostRing(rank, size, "127.0.0.1", port, timeout=60.0)
    big = bytes([rank]) * 300000  # larger than a socket buffer: the sends must not wait for the receives
    from_prev, from_next = ring.sendrecv(b"P%d" % rank + big, b"N%d" % rank + big)
    parts = ring.all_gather(b"G%d" % rank)
    root = ring.broadcast(b"ROOT" if rank == 1 else None, 1)
    ring.barrier()
    ring.close()
    q.put((rank, from_prev[:2], from_next[:2], len(from_prev), parts, root))


@pytest.mark.parametrize("size", [2, 3])
def test_host_ring_of_several_processes(size):
    """pypwt_amd.comm.HostRing (the TCP ring TiledWavelets uses where ranks share a GPU: the tests of the tiled path): neighbour
    exchange, all-gather and broadcast between `size` processes on this machine -- no GPU involved."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_host_ring_rank, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    for r, fp, fn, n, parts, root in res:
        assert fp == b"N%d" % ((r - 1) % size) and fn == b"P%d" % ((r + 1) % size) and n == 300002
        assert parts == [b"G%d" % k for k in range(size)] and root == b"ROOT"


def test_dispatch_thresholds_are_a_table_with_provenance():
    """the dispatch literals of build_schedule and the launchers are rows of csrc/tuning_gfx950.inc -- key,
    value, evidence file, round -- read through tuning.hpp; every evidence file exists, no key twice, and the sources hold no
    bare size threshold of their own where a row exists."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tuning_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tuning_table
    rows = tuning_table.rows()
    assert len(rows) >= 40
    keys = {k for _, k, _, _, _, _ in rows}
    for src in ("plan.cpp", "launch_dwt2.hip", "launch_dwt2_fast.hip", "launch_swt_split.hip"):
        txt = open(os.path.join(ROOT, "pypwt_amd", "csrc", src)).read()
        assert '#include "tuning.hpp"' in txt, src
        used = set(re.findall(r"\btune::(\w+)", txt))
        assert used and used <= keys, (src, used - keys)
    every = set()
    for f in os.listdir(os.path.join(ROOT, "pypwt_amd", "csrc")):
        if f.endswith((".cpp", ".hip")):
            every |= set(re.findall(r"\btune::(\w+)", open(os.path.join(ROOT, "pypwt_amd", "csrc", f)).read()))
    assert keys - every <= {"kRows", "kRowCount"} | set(), ("rows nothing reads", sorted(keys - every))
    # the table as text (docs/TUNING.md) is the generator's output for THIS tuning_gfx950.inc
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tuning_table.py")], capture_output=True, text=True).stdout
    doc = open(os.path.join(ROOT, "docs", "TUNING.md")).read()
    assert out.strip() and out.strip() in doc, "docs/TUNING.md is stale: regenerate its table with tools/tuning_table.py"
