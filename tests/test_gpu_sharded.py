"""ShardedBatch (pypwt_amd/sharded.py) on the GPU box, and the fields of the bench line that only exist with a GPU.

A one-GPU box cannot spread shards over devices; `devices=[0, 0]` builds two plans with their own streams and host
threads on GPU 0, which exercises everything but the second device: the partition, the per-shard input offsets, the
routing of `coeff_at` / `image_at`, concurrent calls from two threads into the library."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("batch,devices", [(5, [0, 0]), (4, [0, 0, 0]), (2, [0, 0, 0])])
def test_sharded_batch_against_the_oracle(batch, devices):
    from pypwt_amd import ShardedBatch
    oracle.build()
    Nr, Nc, wname, L = 256, 192, "db4", 3
    S = ShardedBatch(batch, Nr, Nc, wname, L, devices=devices)
    assert [hi - lo for _, lo, hi in S.shards] == [n for n in
                                                   [batch // len(devices) + (1 if r < batch % len(devices) else 0)
                                                    for r in range(len(devices))] if n > 0]
    S.fill_hash(77, 255.0)
    S.forward()
    per = Nr * Nc
    for b in sorted({0, batch - 1, S.shards[0][2] - 1, S.shards[-1][1]}):  # first / last image of the first and last shard
        x = oracle.hash_input((Nr, Nc), 77, scale=255.0, index_offset=b * per)
        ref = oracle.forward(x, wname, L)
        for num, r in enumerate(ref):
            g = S.coeff_at(num, b)
            assert g.shape == r.shape and np.abs(g - r).max() <= 1.5e-6 * (L + 1) * max(np.abs(r).max(), 1.0), (b, num)
    S.soft_threshold(3.0)
    S.inverse()
    S.synchronize()
    for b in (0, batch - 1):
        x = oracle.hash_input((Nr, Nc), 77, scale=255.0, index_offset=b * per)
        thr = oracle.threshold(oracle.forward(x, wname, L), (Nr, Nc), L, "soft", 3.0)
        rec = oracle.inverse(thr, (Nr, Nc), wname, L)
        assert np.abs(S.image_at(b) - rec).max() <= 3e-6 * 255 * (L + 1), b
    with pytest.raises(IndexError):
        S.image_at(batch)
    S.cleanup()


def test_sharded_set_image_routes_blocks_to_their_owners():
    from pypwt_amd import ShardedBatch
    rng = np.random.RandomState(5)
    x = (rng.rand(3, 64, 128) * 255).astype(np.float32)
    S = ShardedBatch(3, 64, 128, "haar", 2, devices=[0, 0])
    S.set_image(x)
    S.forward()
    S.inverse()
    for b in range(3):
        assert np.abs(S.image_at(b) - x[b]).max() < 7e-4
    S.cleanup()


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, env=e,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_names_the_longest_kernel_and_carries_the_measured_ceiling():
    """VERDICT round 3, task 6: `roofline.kernel` is chosen from launches re-timed ALONE (not from event-inflated in-step
    times), the line carries a flat copy of the same footprint timed in the same run, and the step's copy floor."""
    for attempt in range(2):
        out = _bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--preheat-ms", "100")
        rf, e2e = out["roofline"], out["end_to_end"]
        # the copy floor is timed once per line: on a box whose other GPU slots are busy a reading can come out above the step it
        # bounds (seen once in ten evidence runs: 82.9 us against 53-54) -- one re-measurement, then the assertions below decide
        if e2e["copy_floor_us_per_step"] < out["ms_per_step"] * 1e3 * 1.05:
            break
    assert out["config"]["workload"].startswith("cfg2") and out["n_gpus"] == 1
    timed = [k for k in out["kernels"] if "isolated_us" in k]
    assert len(timed) >= 2
    # the longest of the re-timed candidates -- each judged by its isolated timing unless that is more than 10 % shorter than what the
    # launch takes inside the step (round 6: bench.judged_duration; cfg4's fused inverse is the case it exists for)
    import bench
    assert rf["avg_us"] == max(bench.judged_duration(k)[0] for k in timed)
    assert rf["avg_us_basis"] in ("isolated", "in_step") and rf["isolated_us"] is not None and rf["in_step_us"] is not None
    assert out["target"]["north_star_frac"] == 0.70 and out["target"]["met"] == (out["target"]["end_to_end_frac"] >= 0.70)
    assert abs(out["target"]["end_to_end_frac"] - e2e["frac_of_hbm_peak"]) < 1e-12
    assert rf["kernel"] in ("dwt2_inv_level[L1]", "dwt2_fwd_level[L1]")
    assert 0.3 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-9
    assert rf["copy_ceiling_GBps"] > 3000 and rf["copy_ceiling_bytes"] == rf["algorithmic_bytes_per_launch"]
    assert 0.5 < rf["frac_of_copy_ceiling"] < 1.3
    assert 0 < e2e["copy_floor_us_per_step"] < out["ms_per_step"] * 1e3 * 1.05
    assert abs(e2e["copy_floor_over_step"] - e2e["copy_floor_us_per_step"] * 1e-3 / out["ms_per_step"]) < 1e-6


def test_bench_single_process_sharded_over_the_one_gpu():
    out = _bench("--gpus", "2", "--single-process", "--config", "cfg1", "--batch", "3", "--steps", "5", "--warmup", "2",
                 "--no-cpu-baseline", "--preheat-ms", "20", env={"PDWT_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["config"]["images_per_step"] == 6
    assert out["config"]["shards"] == [[0, 0, 3], [0, 3, 6]] and "ONE process" in out["config"]["parallelism"]
    assert out["value"] > 0 and "shared_gpu_test_run" in out["config"]


def test_bench_default_line_carries_every_baseline_configuration():
    """VERDICT round 4, task 2: the driver's own invocation (no flags but steps / warmup) puts all five BASELINE
    configurations on ONE line -- cfg2 as the headline, cfg1 / cfg3 / cfg4 under extra.configs with their step time, roofline
    fraction, dominant kernel and copy floor, cfg5's shard under extra.cfg5_shard_one_gpu -- within seconds."""
    import time
    t0 = time.time()
    out = _bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    wall = time.time() - t0
    assert "cfg2" in out["metric"] and len(out["config"]["timed_regions_ms"]) == 3
    cfgs = out["extra"]["configs"]
    for name in ("cfg1", "cfg3", "cfg4"):
        c = cfgs[name]
        assert "error" not in c, c
        assert c["steps"] == 20 and c["ms_per_step"] > 0 and 0 < c["frac_of_hbm_peak"] < 1.0, (name, c)
        assert c["dominant_kernel"] and c["dominant_kernel_us"] > 0 and c["copy_floor_us_per_step"] > 0, (name, c)
        assert len(c["timed_regions_ms_per_step"]) == 3
    assert "cfg5_shard_one_gpu" in out["extra"]
    assert wall < 60, wall  # python start-up + plan creation included; the measurements themselves are a few seconds
