"""A single image split in row slabs over several ranks (pypwt_amd/tiled.py): the slabs of every
sub-band and of the reconstruction must equal those of the CPU oracle's transform of the whole image;
levels whose slabs would be thinner than the halo are gathered on rank 0 and compared whole.
The GPU box has ONE GPU: the ranks share it and exchange their halos through pypwt_amd.comm.HostRing (TCP, staged on the
host: RCCL refuses two ranks on one GPU); the library's RCCL transport runs with one rank as its own neighbour.  No torch:
every worker asserts that it never entered sys.modules (round 5)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, wname, levels, shape, swt=0, backend="ring", fuse=0):
    """Each rank is a child process: tests/tiled_worker.py compares its slabs with the oracle itself."""
    port, ring_port = _free_port(), _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PDWT_COMM_NONCE="tiled%d" % port, PDWT_RING_PORT=str(ring_port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tiled_worker.py"), wname,
                                       str(levels), str(shape[0]), str(shape[1]), backend, str(swt), str(fuse)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("OK %d" % r) in out, "rank %d failed:\n%s" % (r, out[-3000:])
    return outs


@pytest.mark.parametrize("wname,levels,shape", [("haar", 3, (64, 96)), ("db2", 2, (64, 64)), ("db4", 3, (192, 160)),
                                                ("sym8", 2, (128, 256)), ("bior2.4", 2, (96, 64))])
def test_single_rank_ring_equals_plain_transform(wname, levels, shape):
    """world size 1, no process group: the ring closes on itself, so the tiled path must reproduce the
    ordinary transform."""
    _run_ranks(1, wname, levels, shape)


@pytest.mark.parametrize("world,wname,levels,shape", [(2, "db4", 3, (256, 128)), (3, "sym4", 2, (192, 64)),
                                                      (4, "haar", 2, (64, 64)), (2, "db8", 2, (256, 96))])
def test_row_slabs_over_ranks_with_halo_exchange(world, wname, levels, shape):
    _run_ranks(world, wname, levels, shape)


@pytest.mark.parametrize("world,wname,levels,shape", [(2, "db2", 4, (80, 64)), (4, "haar", 4, (96, 64)),
                                                      (3, "db3", 4, (120, 96)), (1, "db2", 4, (40, 64)),
                                                      (2, "db2", 5, (160, 128))])
def test_deep_levels_are_gathered_on_rank0(world, wname, levels, shape):
    """more levels than the slabs support: the remaining approximation is gathered (one all-gather), finished on
    rank 0, and handed back by the inverse (one broadcast)"""
    _run_ranks(world, wname, levels, shape)


@pytest.mark.parametrize("world,wname,levels,shape", [(1, "haar", 3, (64, 96)), (2, "haar", 4, (128, 256)),
                                                      (2, "db2", 3, (128, 100)), (3, "sym4", 2, (96, 64)),
                                                      (4, "haar", 5, (512, 256)), (2, "bior2.2", 2, (90, 64))])
def test_swt_slabs_with_one_halo_exchange(world, wname, levels, shape):
    """undecimated transform: ONE exchange of hlen (2^levels - 1) rows per side, the whole multi-level SWT plan on the
    extended slab, interiors kept; every band's slab and the reconstruction against the oracle"""
    _run_ranks(world, wname, levels, shape, swt=1)


@pytest.mark.parametrize("wname,levels,shape,swt", [("db4", 3, (192, 160), 0), ("db2", 4, (40, 64), 0), ("sym8", 2, (128, 256), 0),
                                                    ("haar", 3, (64, 96), 1), ("db2", 3, (128, 100), 1)])
def test_library_rccl_transport_with_one_rank_as_its_own_neighbour(wname, levels, shape, swt):
    """The library's OWN transport (pdwt_comm_*, pypwt_amd/comm.py: RCCL dlopen'ed by the C library, grouped send / recv on
    the plans' stream, all_gather + broadcast for the gathered levels), one rank as its own neighbour: every slab of every
    band and the reconstruction against the CPU oracle."""
    outs = _run_ranks(1, wname, levels, shape, swt=swt, backend="comm")
    assert "comm-loopback" in outs[0]


@pytest.mark.parametrize("world,wname,levels,shape,backend,want", [
    (1, "db4", 3, (1024, 160), "none", "1+3"), (2, "db4", 4, (2048, 128), "ring", "1+1,2+1,3+2"), (1, "sym8", 3, (2048, 256), "comm", "1+3"),
    (4, "db2", 3, (1024, 64), "ring", "1+1,2+2"), (2, "haar", 3, (64, 64), "ring", "1+3"), (1, "db8", 2, (1024, 96), "comm", "1+2")])
def test_last_slab_levels_run_as_one_group(world, wname, levels, shape, backend, want):
    """Round 5: the last K slab levels are ONE K-level plan behind ONE exchange per direction (hp (2^K - 1) image rows forward; the
    inverse: q_j rows of the j-th level's bands, q_1 = hq, q_(j+1) = hq + ceil(q_j / 2)) -- the worker reports its level groups;
    the same cases with at most two levels per group and with every level its own group must give the same slabs."""
    outs = _run_ranks(world, wname, levels, shape, backend=backend)
    assert outs[0].split("groups=")[1].split()[0] == want, outs[0][-300:]
    outs = _run_ranks(world, wname, levels, shape, backend=backend, fuse=2)
    assert outs[0].split("groups=")[1].split()[0].endswith("+2"), outs[0][-300:]
    outs = _run_ranks(world, wname, levels, shape, backend=backend, fuse=1)
    assert outs[0].split("groups=")[1].split()[0].endswith("+1"), outs[0][-300:]


def test_communicator_without_torch():
    """pypwt_amd.comm.Communicator in a process that never imports torch: a ring of one exchanges rows between two plans'
    buffers (send to self / receive from self in ONE group), gathers and broadcasts, all on a plan's stream."""
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
from pypwt_amd import BatchedWavelets
from pypwt_amd.comm import Communicator
c = Communicator.single()
assert (c.rank, c.size) == (0, 1)
x = (np.arange(64 * 96, dtype=np.float32).reshape(1, 64, 96) %% 251) - 100
A = BatchedWavelets(1, 64, 96, "haar", 1, img=x)
B = BatchedWavelets(1, 64, 96, "haar", 1)
import ctypes as C
lib = A._lib
s = lib.pdwt_get_stream(A._h)
B.set_stream(s)                     # both plans on one stream: the transfers are ordered with their kernels
pa, pb = lib.pdwt_image_ptr(A._h), lib.pdwt_image_ptr(B._h)
row = 96 * 4
# rows 8..15 of A -> rows 0..7 of B, rows 40..47 of A -> rows 56..63 of B: two messages in one group
c.exchange([(pa + 8 * row, 8 * 96, 0), (pa + 40 * row, 8 * 96, 0)], [(pb, 8 * 96, 0), (pb + 56 * row, 8 * 96, 0)], stream=s)
c.all_gather(pa + 16 * row, pb + 16 * row, 4 * 96, stream=s)
c.broadcast(pb + 32 * row, 96, 0, stream=s)
A.synchronize()
got = B.image_at(0)
assert np.array_equal(got[0:8], x[0, 8:16]) and np.array_equal(got[56:64], x[0, 40:48]), "exchange"
assert np.array_equal(got[16:20], x[0, 16:20]), "all_gather"
assert not got[8:16].any() and not got[20:56].any()
c.close()
assert "torch" not in sys.modules
print("COMM-OK")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "COMM-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_tiled_path_does_not_import_torch():
    """pypwt_amd.tiled in a process that never imports torch (round 5: DeviceRows views, pdwt_copy, Communicator / HostRing)."""
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from pypwt_amd.tiled import TiledWavelets\n"
            "x = (np.arange(128 * 64, dtype=np.float32).reshape(128, 64) %% 97)\n"
            "t = TiledWavelets(x, 'db2', 3)\n"
            "t.forward(); t.inverse()\n"
            "assert np.abs(t.image - x).max() < 1e-3 and 'torch' not in sys.modules\n"
            "print('NO-TORCH-OK', t.groups)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "NO-TORCH-OK" in out.stdout, out.stdout + out.stderr[-3000:]


@pytest.mark.parametrize("config,extra", [("cfg1", []), ("cfg2", ["--batch", "2", "--scaling", "strong"])])
def test_bench_two_ranks_on_the_gpu_box(config, extra):
    """`python bench.py --gpus 2` on real hardware as far as a one-GPU box allows: the parent starts two ranks, torch is
    imported before libpypwt_amd.so in each, both run their shard on GPU 0 (PDWT_BENCH_SHARE_GPU=1) behind gloo barriers,
    rank 0 prints ONE JSON line with the aggregate over both ranks.  (RCCL itself runs in bench.py --force-dist with one
    rank: profiles/r03z_bench_cfg2_nccl_world1.json.)"""
    import json
    env = dict(os.environ, PDWT_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--config", config,
                        "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--preheat-ms", "20"] + extra,
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["value"] > 0
    assert "shared_gpu_test_run" in out["config"] and "multi_gpu_note" in out["config"]
    assert out["config"]["images_per_step"] == 2 and out["scaling"] == ("strong" if extra else "weak")
    assert "cpu_baseline" not in out or out["cpu_baseline"] is None or isinstance(out["cpu_baseline"], dict)
    # the line carries its own one-GPU reference on the SAME workload (rank 0 alone on its shard) and three timed regions
    ref = out["scaling_reference"]
    assert ref["one_gpu_same_workload_Msamples_s"] > 0 and ref["one_gpu_same_workload_ms_per_step"] > 0
    assert abs(ref["efficiency"] - out["value"] / (2 * ref["one_gpu_same_workload_Msamples_s"])) < 1e-6 * max(1.0, ref["efficiency"])
    assert 0.2 < ref["efficiency"] < 1.3, ref  # two ranks SHARING one GPU: about half of perfect scaling, never more than all of it
    assert len(out["config"]["timed_regions_ms"]) == 3 and config in out["metric"]
