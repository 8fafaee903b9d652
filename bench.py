#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native wavelet transform.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Msamples/s + %HBM-roofline on 4096x4096 fp32 db4 4-level 2D DWT
forward+inverse.  A "step" is one forward + one inverse of `--batch` images per GPU with the input
already resident in HBM (generated on the device).  With N > 1 every rank owns an independent plan on
its own GPU and transforms its own images: the path shards by image, there is NO collective on the
data path; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the time.
K steps x batch B per rank is exactly BASELINE config 5's "batch of images sharded over the GPUs"
(1024 images over 8 GPUs = --steps 128 --batch 1, or --steps 1 --batch 128).

Rank 0 prints ONE JSON line: the contract's keys plus
  roofline      dominant kernel: algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline  the C oracle (a port of the reference's algorithm) timed on this host's cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

CONFIGS = {
    # name: (Nr, Nc, wname, levels, do_swt, ndim, soft-threshold beta or None, description)
    "cfg2": (4096, 4096, "db4", 4, 0, 2, None, "4096x4096 fp32 db4 4-level separable 2D DWT fwd+inv"),
    "cfg3": (1, 1 << 24, "sym8", 6, 0, 1, None, "1D 2^24-sample fp32 sym8 6-level DWT fwd+inv"),
    "cfg4": (2048, 2048, "haar", 5, 1, 2, 25.5, "SWT 2048x2048 fp32 haar 5-level fwd + soft_threshold + inv"),
    "cfg1": (512, 512, "db2", 3, 0, 2, None, "512x512 fp32 db2 3-level 2D DWT fwd+inv"),
}


def algorithmic_bytes_per_sample(cfg):
    """Compulsory HBM traffic per input sample of one step (SURVEY.md 8d / BASELINE.md 4)."""
    Nr, Nc, wname, L, swt, ndim, beta, _ = cfg
    if not swt:
        return 16.0  # fwd: read x, write all coefficients; inv: read them, write x
    nb = 3 * L + 1 if ndim == 2 else L + 1
    b = 4.0 * (1 + nb) + 4.0 * (nb + 1)
    if beta is not None:
        b += 2 * 4.0 * (nb - 1)
    return b


def kernel_algorithmic_bytes(label, cfg, batch):
    """Algorithmic bytes of ONE launch: its input read once + its outputs written once (DESIGN.md
    'Kernels').  label = 'name[Ll]' from label_step_kernels."""
    Nr, Nc, wname, L, swt, ndim, beta, _ = cfg
    samples = batch * Nr * Nc
    name, lvl = label[:-1].split("[")
    name = name.replace("+soft", "")  # SWT inverse level with the deferred soft-threshold folded in
    lvl = None if lvl == "-" else int(lvl[1:])
    if name in ("dwt1_fwd_fused", "dwt1_inv_fused"):
        return 8.0 * samples / (2 ** (lvl - 1))  # all remaining levels: input once, every band once
    if name in ("dwt2_fwd_pyr2", "dwt2_inv_pyr2", "dwt2_fwd_strip2", "dwt2_inv_strip2"):
        # two levels in one launch: the pair's input read once, details of the first and all four bands
        # of the second level written once = 8 B per sample entering the pair
        return 8.0 * samples / (4 ** (lvl - 1))
    if name.startswith("dwt2"):
        return 8.0 * samples / (4 ** (lvl - 1))
    if name.startswith("dwt1"):
        return 8.0 * samples / (2 ** (lvl - 1))
    if name in ("swt2_fwd_level", "swt2_inv_level"):
        return 4.0 * 5 * samples
    if name in ("swt1_fwd_level", "swt1_inv_level"):
        return 4.0 * 3 * samples
    if name == "soft_threshold":
        nb = 3 * L if ndim == 2 else L
        return 8.0 * nb * samples
    return 0.0


def label_step_kernels(names, L):
    """Launch names of ONE step, in order -> labels 'name[Ll]' with l = the (first) transform level the
    launch works on.  Forward launches count levels up from 1, inverse launches down from L; a pyramid
    launch covers two levels, a fused 1D launch all remaining ones."""
    out = []
    f, i = 1, L
    for n in names:
        base = n.replace("+soft", "")
        if base in ("dwt2_fwd_level", "dwt1_fwd_level", "swt2_fwd_level", "swt1_fwd_level", "nonsep_fwd_level"):
            out.append("%s[L%d]" % (n, f)); f += 1
        elif base in ("dwt2_fwd_pyr2", "dwt2_fwd_strip2"):
            out.append("%s[L%d]" % (n, f)); f += 2
        elif base == "dwt1_fwd_fused":
            out.append("%s[L%d]" % (n, f)); f = L + 1
        elif base in ("dwt2_inv_level", "dwt1_inv_level", "swt2_inv_level", "swt1_inv_level", "nonsep_inv_level"):
            out.append("%s[L%d]" % (n, i)); i -= 1
        elif base in ("dwt2_inv_pyr2", "dwt2_inv_strip2"):
            out.append("%s[L%d]" % (n, i - 1)); i -= 2
        elif base == "dwt1_inv_fused":
            out.append("%s[L%d]" % (n, 1)); i = 0
        else:
            out.append("%s[-]" % n)
    return out


def level_of_kernel(kernel, L):
    """'dwt2_inv_level[L1]' -> (level, is_inverse) for pdwt_time_level; None when the launch is not a
    single-level one that pdwt_time_level can repeat."""
    name, lvl = kernel[:-1].split("[")
    if lvl == "-":
        return None
    return int(lvl[1:]), ("_inv_" in name)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=200.0,
                    help="run the workload untimed for this long before the warmup steps: after an idle period "
                         "the GPU needs tens of ms to ramp its clocks, and W = 10 steps last under 1 ms")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU/gloo plumbing test: no GPU work, exercises rendezvous + aggregation only")
    return ap.parse_args()


def init_dist(args):
    """One process per GPU.  Returns (rank, world, local_rank, dist or None, backend).

    torch is imported ONLY when WORLD_SIZE > 1 (barrier + max-over-ranks); the single-GPU run
    never loads it, so the HIP library runs on the ROCm runtime it was built with.  (torch wheels
    bundle their own libamdhip64 with the same soname; in a multi-rank run torch is loaded first and
    the library shares that runtime.)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, 0, None, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    backend = "gloo" if args.dry_run else args.dist_backend
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank, dist, backend


def barrier(dist, backend):
    if dist is None:
        return
    if backend == "nccl":
        import torch
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def max_over_ranks(value, dist, backend):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_images(total, world, rank):
    """Images [lo, hi) of a batch of `total` owned by `rank` (contiguous blocks, remainder to the
    first ranks).  Used for the deterministic per-rank input offset; no data is exchanged."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def cpu_baseline(cfg):
    """The C oracle (a port of the reference's algorithm, fp32, OpenMP over rows) on this host:
    a bounded sample of the same workload -- one image, forward + inverse, best of 3."""
    import numpy as np

    from oracle import oracle
    oracle.build()
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    cores = oracle.set_threads(oracle.usable_cpus())
    x = oracle.hash_input((Nr, Nc), 20242)
    best = None
    reps = 3 if not swt else 1
    for _ in range(reps):
        t0 = time.perf_counter()
        bands = oracle.forward(x, wname, L, ndim=ndim, do_swt=swt)
        if beta is not None:
            bands = oracle.threshold(bands, (Nr, Nc), L, "soft", beta, ndim=ndim, do_swt=swt)
        rec = oracle.inverse(bands, (Nr, Nc), wname, L, ndim=ndim, do_swt=swt)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.isfinite(rec).all()
    return {"value": Nr * Nc / best / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "1 image of the bench workload (%s), best of %d, oracle/pdwt_oracle.c fp32 + OpenMP" % (desc, reps),
            "seconds": best}


def dry_run(args, rank, world, dist, backend):
    """No GPU: sleep-based steps so the launch / barrier / max-over-ranks / JSON path is testable
    with gloo on CPU."""
    cfg = CONFIGS[args.config]
    barrier(dist, backend)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    dt = max_over_ranks(time.perf_counter() - t0, dist, backend)
    barrier(dist, backend)
    lo, hi = shard_images(world * args.batch * args.steps, world, rank)
    if rank == 0:
        samples = world * args.batch * cfg[0] * cfg[1]
        print(json.dumps({"metric": "dry_run", "value": samples / (dt / args.steps) / 1e6, "unit": "Msamples/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": "dry-run", "shard_rank0": [lo, hi]}}))


def main():
    args = parse_args()
    rank, world, local_rank, dist, backend = init_dist(args)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if args.dry_run:
        dry_run(args, rank, world, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        return

    from pypwt_amd import BatchedWavelets
    cfg = CONFIGS[args.config]
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    B = args.batch

    plan = BatchedWavelets(B, Nr, Nc, wname, L, do_swt=swt, ndim=ndim, device=local_rank)
    # deterministic synthetic input generated ON the device; every rank gets different images
    lo, _ = shard_images(world * B, world, rank)
    plan.fill_hash(20240 + 2, 255.0, index_offset=lo * Nr * Nc)

    def step():
        plan.forward()
        if beta is not None:
            plan.soft_threshold(beta)
        plan.inverse()

    def device_sync():
        plan.synchronize()  # every kernel of this rank runs on the plan's stream
        if dist is not None and backend == "nccl":
            import torch
            torch.cuda.synchronize()

    # pre-heat (untimed): sustained-throughput conditions for the timed steps
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(20):
            step()
        device_sync()
    for _ in range(args.warmup):
        step()
    device_sync()
    barrier(dist, backend)
    device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_sync()
    barrier(dist, backend)
    device_sync()
    dt = max_over_ranks(time.perf_counter() - t0, dist, backend)

    step_s = dt / args.steps
    samples_per_step = world * B * Nr * Nc
    value = samples_per_step / step_s / 1e6

    # ---- per-kernel durations from HIP events on the plan's stream (second pass, same steps)
    plan.enable_kernel_timing(True)
    plan.reset_kernel_times()
    for _ in range(args.steps):
        step()
    times = plan.kernel_times(cap=64 * args.steps + 64)
    plan.enable_kernel_timing(False)
    plan.reset_kernel_times()
    per_step = len(times) // args.steps
    labels = label_step_kernels([n for n, _ in times[:per_step]], L)
    agg = {}
    for i, (name, ms) in enumerate(times):
        agg.setdefault(labels[i % per_step], []).append(ms)
    kernels = []
    for label, v in agg.items():
        avg_ms = sum(v) / len(v)
        abytes = kernel_algorithmic_bytes(label, cfg, B)
        kernels.append({"kernel": label, "avg_us": avg_ms * 1e3, "algorithmic_bytes": abytes,
                        "GBps": abytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0})
    kernels.sort(key=lambda k: -k["avg_us"])
    dom = kernels[0]
    # The per-launch events above cost ~2.5 us of stream time each (sum of the in-step durations exceeds
    # the step), so the dominant kernel is timed again on its own: `steps` launches of that level back
    # to back between TWO HIP events on the plan's stream (pdwt_time_level).  That figure agrees with
    # rocprofv3 --kernel-trace (profiles/) and is the one the roofline uses.
    dom_us = dom["avg_us"]
    lvl = level_of_kernel(dom["kernel"], L)
    if lvl is not None:
        plan.forward()  # valid data in every buffer the level reads
        dom_us = plan.time_level(lvl[0], inverse=lvl[1], reps=max(args.steps, 20))
        plan.inverse()
    dom_gbps = dom["algorithmic_bytes"] / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": dom_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": dom_gbps / HBM_PEAK_GBPS, "traffic": None, "kernel": dom["kernel"],
                "avg_us": dom_us, "avg_us_in_step_with_event_overhead": dom["avg_us"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes"]}
    # HBM bytes per launch of that kernel from the rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 + WRITE_SIZE, collected separately; tools/prof.sh + tools/summarize_pmc.py).
    # Counters cannot be read from inside the process, so the committed measurement is quoted.
    tpath = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % args.config)
    if B == 1 and os.path.exists(tpath):
        try:
            # the profile labels launches by kernel family: "+soft" (a deferred threshold folded into the SWT
            # inverse) is a property of the plan state, not of the kernel
            t = json.load(open(tpath))["per_launch"].get(dom["kernel"].replace("+soft", ""))
            if t:
                roofline["traffic"] = t["hbm_bytes"]
                roofline["traffic_source"] = "profiles/" + os.path.basename(tpath)
        except Exception:
            pass
    e2e_bytes = algorithmic_bytes_per_sample(cfg) * B * Nr * Nc  # per GPU per step
    e2e = {"algorithmic_bytes_per_step_per_gpu": e2e_bytes, "GBps_per_gpu": e2e_bytes / step_s / 1e9,
           "frac_of_hbm_peak": e2e_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
           "sum_kernel_us": sum(k["avg_us"] for k in kernels)}

    out = {
        "metric": "Msamples/s, 4096x4096 fp32 db4 L4 2D DWT fwd+inv" if args.config == "cfg2" else "Msamples/s, " + desc,
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (on-device index hash, 0..255)",
        "config": {"workload": "%s: %s" % (args.config, desc), "batch_per_gpu": B, "wavelet": wname, "levels": L,
                   "shape": [Nr, Nc], "parallelism": "image-sharded x%d, no collectives" % world,
                   "preheat_ms": args.preheat_ms},
        "roofline": roofline, "end_to_end": e2e, "kernels": kernels[:12],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(out))
    plan.cleanup()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
