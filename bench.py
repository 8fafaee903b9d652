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
`python bench.py --gpus N` without a launcher starts the N ranks itself (one child process per GPU,
before anything touches a GPU).  `--config cfg5` is BASELINE config 5's per-GPU shard: 128 images of
4096^2 per GPU and step (1024 images over 8 GPUs).  `--scaling strong` fixes the TOTAL batch (--batch
images split over the ranks) instead of the per-GPU batch.

Rank 0 prints ONE JSON line: the contract's keys plus
  roofline      dominant kernel: algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline  the C oracle (a port of the reference's algorithm) timed on this host's cores
  end_to_end    algorithmic bytes of the whole step / step time, cold (no pre-heat) step time
  extra         (1 GPU) one-thread CPU baseline, quoted / measured pywt, PCIe transfers and the reference's
                own benchmark method (set_image + forward + coeffs), a working set beyond the Infinity Cache
"""
import argparse
import json
import os
import socket
import subprocess
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
    "cfg5": (4096, 4096, "db4", 4, 0, 2, None, "batch of 4096x4096 fp32 db4 4-level 2D DWT fwd+inv, 128 images per GPU"),
}
DEFAULT_BATCH = {"cfg5": 128}

# pywt 1.1.1 timed in the BUILD container (one core of an 8-vCPU Xeon 2.1 GHz; pywt is single-threaded),
# quoted when pywt is not importable on the GPU box
PYWT_QUOTED = {"cfg1": 32.6, "cfg2": 12.6, "cfg3": 24.3, "cfg4": 1.16, "cfg5": 12.6}


def algorithmic_bytes_per_sample(cfg, threshold_separate=True):
    """Compulsory HBM traffic per input sample of one step (SURVEY.md 8d / BASELINE.md 4): input read once,
    final coefficients written once, coefficients read once, output written once.  An SWT soft_threshold
    costs a read + write of every detail plane ONLY when it runs as its own sweep (`threshold_separate`);
    folded into the inverse's loads (the deferred threshold of 2D SWT plans) it moves nothing."""
    Nr, Nc, wname, L, swt, ndim, beta, _ = cfg
    if not swt:
        return 16.0  # fwd: read x, write all coefficients; inv: read them, write x
    nb = 3 * L + 1 if ndim == 2 else L + 1
    b = 4.0 * (1 + nb) + 4.0 * (nb + 1)
    if beta is not None and threshold_separate:
        b += 2 * 4.0 * (nb - 1)
    return b


def per_level_streaming_bytes_per_sample(cfg, threshold_separate=True):
    """Diagnostic: every level reads its input and writes its outputs once."""
    Nr, Nc, wname, L, swt, ndim, beta, _ = cfg
    if swt:
        planes = 5 if ndim == 2 else 3
        b = 2 * 4.0 * planes * L
        if beta is not None and threshold_separate:
            b += 2 * 4.0 * (planes - 2) * L
        return b
    q = 4.0 if ndim == 2 else 2.0
    return 2 * 8.0 * sum(q ** -l for l in range(L))


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
    if name in ("dwt2_fwd_pyr2", "dwt2_inv_pyr2", "dwt2_fwd_strip2", "dwt2_inv_strip2", "dwt2_fwd_wave2", "dwt2_inv_wave2"):
        # two levels in one launch: the pair's input read once, details of the first and all four bands
        # of the second level written once = 8 B per sample entering the pair
        return 8.0 * samples / (4 ** (lvl - 1))
    if name in ("dwt2_fwd_chain", "dwt2_inv_chain"):  # K levels in one launch with in-launch hand-offs: the group's input
        # and final coefficients once = 8 B per sample entering it (the A_l hand-offs inside are not algorithmic bytes)
        return 8.0 * samples / (4 ** (lvl - 1))
    if name in ("dwt2_fwd_pyr3", "dwt2_inv_pyr3"):  # three levels in one launch: 8 B per sample entering the group
        return 8.0 * samples / (4 ** (lvl - 1))
    if name.startswith("dwt2"):
        return 8.0 * samples / (4 ** (lvl - 1))
    if name.startswith("dwt1"):
        return 8.0 * samples / (2 ** (lvl - 1))
    if name in ("swt2_fwd_level", "swt2_inv_level", "swt2_fwd_split", "swt2_inv_split"):
        return 4.0 * 5 * samples
    if name in ("swt2_fwd_fused", "swt2_inv_fused"):  # K levels: one plane in, 3 K + 1 planes out (or the reverse)
        K = min(3, L - lvl + 1)
        return 4.0 * (3 * K + 2) * samples
    if name in ("swt1_fwd_level", "swt1_inv_level"):
        return 4.0 * 3 * samples
    if name == "soft_threshold":
        nb = 3 * L if ndim == 2 else L
        return 8.0 * nb * samples
    return 0.0


def chain_end(first, L, chain_levels=None):
    """Last level of a chain launch that starts at level `first` (chain_levels = K when known, else everything left)."""
    return min(L, first + (chain_levels if chain_levels else L) - 1)


def label_step_kernels(names, L, chain_levels=None):
    """Launch names of ONE step, in order -> labels 'name[Ll]' with l = the (first) transform level the
    launch works on.  Forward launches count levels up from 1, inverse launches down from L; a pyramid
    launch covers two levels, a fused 1D launch all remaining ones."""
    out = []
    f, i = 1, L
    for n in names:
        base = n.replace("+soft", "")
        if base in ("dwt2_fwd_level", "dwt2_fwd_split", "dwt1_fwd_level", "swt2_fwd_level", "swt2_fwd_split", "swt1_fwd_level",
                    "nonsep_fwd_level"):
            out.append("%s[L%d]" % (n, f)); f += 1
        elif base in ("dwt2_fwd_pyr2", "dwt2_fwd_strip2", "dwt2_fwd_wave2"):
            out.append("%s[L%d]" % (n, f)); f += 2
        elif base == "dwt2_fwd_pyr3":
            out.append("%s[L%d]" % (n, f)); f += 3
        elif base == "dwt2_fwd_tail":  # all remaining levels of a small approximation in one launch
            out.append("%s[L%d]" % (n, f)); f = L + 1
        elif base == "dwt2_fwd_chain":  # levels f .. min(L, f + 5) whose tiles are whole (plan.cpp: chain_at); all of cfg2's
            out.append("%s[L%d]" % (n, f)); f = chain_end(f, L, chain_levels) + 1
        elif base == "dwt1_fwd_fused":
            out.append("%s[L%d]" % (n, f)); f = L + 1
        elif base == "swt2_fwd_fused":  # levels 1-3 / 4-6 of a 2-tap SWT in one launch (two levels when only two are left)
            out.append("%s[L%d]" % (n, f)); f += min(3, L - f + 1)
        elif base == "dwt1_fwd_reg":  # up to three levels per launch
            out.append("%s[L%d]" % (n, f)); f = min(f + 3, L + 1)
        elif base in ("dwt2_inv_level", "dwt2_inv_split", "dwt1_inv_level", "swt2_inv_level", "swt2_inv_split", "swt1_inv_level",
                      "nonsep_inv_level"):
            out.append("%s[L%d]" % (n, i)); i -= 1
        elif base in ("dwt2_inv_pyr2", "dwt2_inv_strip2", "dwt2_inv_wave2"):
            out.append("%s[L%d]" % (n, i - 1)); i -= 2
        elif base == "dwt2_inv_pyr3":
            out.append("%s[L%d]" % (n, i - 2)); i -= 3
        elif base == "dwt2_inv_chain":  # always starts at level 1
            out.append("%s[L%d]" % (n, 1)); i = 0
        elif base == "dwt1_inv_fused":
            out.append("%s[L%d]" % (n, 1)); i = 0
        elif base == "swt2_inv_fused":  # groups start at levels 1 and 4
            first = 4 if i >= 4 else 1
            out.append("%s[L%d]" % (n, first)); i = first - 1
        elif base == "dwt1_inv_reg":  # the chunks are cut from level 1 upwards: [1-3], [4-6], ...; undone last first
            first = 3 * ((i - 1) // 3) + 1
            out.append("%s[L%d]" % (n, first)); i = first - 1
        else:
            out.append("%s[-]" % n)
    return out


def labels_from_schedule(names, schedule):
    """Launch names of ONE step + the plan's schedule text ("fwd: REG1D[1-3] FUSED1D[4-6]\ninv: ...", pdwt_schedule_string)
    -> labels 'name[Ll]' with l = the first level of the schedule step each launch belongs to.  One launch list entry per
    schedule step, in order (operator launches such as soft_threshold are not steps); None when the counts do not match
    (a step that fell back to a launch per level) -- the caller then labels by counting (label_step_kernels)."""
    import re
    steps = {"fwd": [], "inv": []}
    for line in schedule.splitlines():
        d = line.split(":", 1)[0].strip()
        if d in steps:
            steps[d] = [int(m) for m in re.findall(r"[A-Z0-9]+\[(\d+)", line)]
    out, fi, ii = [], 0, 0
    for n in names:
        if "_fwd_" in n:
            if fi >= len(steps["fwd"]):
                return None
            out.append("%s[L%d]" % (n, steps["fwd"][fi])); fi += 1
        elif "_inv_" in n:
            if ii >= len(steps["inv"]):
                return None
            out.append("%s[L%d]" % (n, steps["inv"][ii])); ii += 1
        else:
            out.append("%s[-]" % n)
    return out if (fi == len(steps["fwd"]) and ii == len(steps["inv"])) else None


def level_of_kernel(kernel, L):
    """'dwt2_inv_level[L1]' -> (level, is_inverse) for pdwt_time_level; None when the launch is not a
    single-level one that pdwt_time_level can repeat."""
    name, lvl = kernel[:-1].split("[")
    if lvl == "-":
        return None
    return int(lvl[1:]), ("_inv_" in name)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU per step (weak scaling, default 1; cfg5: 128) or in total (--scaling strong)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: cfg2 (one 4096^2 image per step) on one GPU; cfg5 (BASELINE config 5's per-GPU shard, 128 "
                         "images of 4096^2 per GPU and step) with more than one rank, where a 70-us step is too short an "
                         "interval to compare ranks on")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU; strong: --batch images in total, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extra` measurements (one-thread CPU, pywt, PCIe, beyond-Infinity-Cache batch)")
    ap.add_argument("--preheat-ms", type=float, default=200.0,
                    help="run the workload untimed for this long before the warmup steps: after an idle period "
                         "the GPU needs tens of ms to ramp its clocks, and W = 10 steps last under 1 ms")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus devices through pypwt_amd.ShardedBatch (contiguous image blocks, "
                         "one plan + stream + host thread per GPU, no collective, no torch) instead of one process per GPU")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU/gloo plumbing test: no GPU work, exercises rendezvous + aggregation only")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for WORLD_SIZE = 1 (exercises the RCCL barrier / "
                         "all-reduce code and torch-HIP next to libpypwt_amd.so on a one-GPU box)")
    ap.add_argument("--dry-run-barrier-skew-ms", type=float, default=0.0,
                    help="(dry run, tests) rank r enters the CLOSING barrier r x this many ms late: the reported "
                         "time must not change, the barrier is outside the timed interval")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks here, one child process per GPU, with the
    same environment torch.distributed.run would give them.  This process never touches a GPU (no HIP call,
    no torch import) -- it only waits and forwards rank 0's JSON line."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: ranks failed (rank, exit code): %s" % bad, file=sys.stderr)
        return max(abs(rc) for _, rc in bad) or 1
    return 0


def init_dist(args):
    """One process per GPU.  Returns (rank, world, local_rank, dist or None, backend).

    torch is imported ONLY when WORLD_SIZE > 1 (barrier + max-over-ranks); the single-GPU run
    never loads it, so the HIP library runs on the ROCm runtime it was built with.  (torch wheels
    bundle their own libamdhip64 with the same soname; in a multi-rank run torch is loaded first and
    the library shares that runtime.)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not args.force_dist:
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


def metric_name(config, desc, batch_per_gpu):
    """The line's `metric`: BASELINE.json's metric for the db4 configurations, WITH the workload it was measured on (one image
    per step / a batch per GPU): lines of different workloads must not be mistaken for each other."""
    if config == "cfg2":
        return "Msamples/s, 4096x4096 fp32 db4 L4 2D DWT fwd+inv, %s (cfg2)" % ("one image per step" if batch_per_gpu == 1 else "%d images per step" % batch_per_gpu)
    if config == "cfg5":
        return "Msamples/s, 4096x4096 fp32 db4 L4 2D DWT fwd+inv, batch of %d images per GPU and step (cfg5)" % batch_per_gpu
    return "Msamples/s, %s (%s)" % (desc, config)


NORTH_STAR_FRAC = 0.70  # BASELINE.json: >= 70 % of the HBM roofline, 4096^2 fp32 db4 L4 forward + inverse, END TO END


def target_record(e2e_frac, algorithmic_bytes_per_step, copy_floor_us):
    """What the line is graded against, next to what it reached: the north-star fraction (end to end, not the dominant
    kernel alone), the end-to-end fraction measured, and the fraction the step would reach if every one of its launches
    were a flat copy of its own bytes (the bound of a one-launch-per-level schedule, measured in this run)."""
    floor = None
    if copy_floor_us and copy_floor_us > 0:
        floor = algorithmic_bytes_per_step / (copy_floor_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
    return {"north_star_frac": NORTH_STAR_FRAC, "of": "end-to-end algorithmic bytes per step / step time / %.0f GB/s" % HBM_PEAK_GBPS,
            "end_to_end_frac": e2e_frac, "launch_copy_floor_frac": floor, "met": bool(e2e_frac >= NORTH_STAR_FRAC)}


def in_step_durations(kernels, step_us):
    """The event pair around every launch of a step costs stream time (the in-step averages sum to more than the step):
    spread the excess evenly over the launches and take it off -- `in_step_us` is what a launch takes INSIDE the pipelined
    step (what rocprofv3 --kernel-trace shows), next to `isolated_us`, the same launch repeated back to back on its own."""
    n = len(kernels)
    total = sum(k["avg_us"] for k in kernels)
    overhead = max(0.0, (total - step_us) / n) if n and step_us else 0.0
    for k in kernels:
        k["in_step_us"] = max(k["avg_us"] - overhead, 0.0)
    return overhead


def judged_duration(k):
    """The duration a launch's roofline fraction is computed from: the isolated timing, unless that is more than 10 % shorter
    than what the launch takes in the step (planes that fit the Infinity Cache only when the launch runs alone: cfg4's fused
    SWT inverse, 35.8 us alone against 42.7 in the step)."""
    iso, ins = k.get("isolated_us"), k.get("in_step_us")
    if iso is None:
        return ins if ins is not None else k["avg_us"], "in_step"
    if ins is not None and iso < 0.9 * ins:
        return ins, "in_step"
    return iso, "isolated"


def median(values):
    v = sorted(values)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def resolve_config(args, world):
    """--config when given; else the headline one-image configuration on one GPU and BASELINE config 5's per-GPU shard
    (the same images, 128 per GPU and step) when the batch is sharded over several: with the driver's --steps 20 a
    one-image step gives a 1.4 ms interval, of which the rank-to-rank start skew behind a barrier is a visible share."""
    if args.config is None:
        args.config = "cfg5" if world > 1 else "cfg2"
        args.config_defaulted = True
    return args.config


def rank_batch(args, world, rank):
    """(images of this rank per step, first image index, total images per step)."""
    resolve_config(args, world)
    b = args.batch if args.batch is not None else DEFAULT_BATCH.get(args.config, 1)
    if args.scaling == "strong":
        lo, hi = shard_images(b, world, rank)
        return hi - lo, lo, b
    return b, rank * b, world * b


def cpu_baseline(cfg, threads=None):
    """The C oracle (a port of the reference's algorithm, fp32, OpenMP over rows) on this host's cores (or on
    `threads` of them): a bounded sample of the same workload -- one image, forward (+ threshold) + inverse,
    best of 3 (one run for the slow configurations).  The ONLY place bench.py touches oracle/."""
    import numpy as np

    from oracle import oracle
    oracle.build()
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    cores = oracle.set_threads(threads if threads else oracle.usable_cpus())
    x = oracle.hash_input((Nr, Nc), 20242)
    best = None
    reps = (3 if not swt else 1) if cores > 1 else 1
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
            "sample": "1 image of the bench workload (%s), best of %d, oracle/pdwt_oracle.c fp32%s"
                      % (desc, reps, " + OpenMP" if cores > 1 else ", one thread"),
            "seconds": best}


def synthetic_host_image(Nr, Nc):
    import numpy as np
    return (np.random.RandomState(0).rand(Nr, Nc) * 255.0).astype(np.float32)


def pywt_baseline(name, cfg):
    """pywt on this host when importable (the reference's own CPU oracle, test/test_wavelets.py:230,301,372,438;
    single-threaded by nature), else the figure measured in the build container, labelled."""
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    quoted = {"value": PYWT_QUOTED[name], "unit": "Msamples/s", "cores": 1,
              "kind": "quoted, build container (pywt 1.1.1, one core of an 8-vCPU Xeon 2.1 GHz; BASELINE.md 2)"}
    try:
        import numpy as np
        import pywt
    except Exception:
        return quoted
    try:
        x = synthetic_host_image(Nr, Nc)
        t0 = time.perf_counter()
        if swt:
            co = pywt.swt2(x, wname, L)
            if beta is not None:
                co = [(a, tuple(pywt.threshold(d, beta, "soft") for d in det)) for a, det in co]
            pywt.iswt2(co, wname)
        elif ndim == 2:
            pywt.waverec2(pywt.wavedec2(x, wname, mode="periodization", level=L), wname, mode="periodization")
        else:
            pywt.waverec(pywt.wavedec(x[0], wname, mode="periodization", level=L), wname, mode="periodization")
        dt = time.perf_counter() - t0
        return {"value": Nr * Nc / dt / 1e6, "unit": "Msamples/s", "cores": 1,
                "kind": "measured here, pywt %s" % pywt.__version__, "seconds": dt}
    except Exception as e:  # never let the optional baseline break the bench line
        quoted["note"] = "pywt import ok but the run failed: %r" % (e,)
        return quoted


def transfers(name, cfg, device):
    """PCIe side of the API, reported separately from `value`: H2D of one image (set_image), D2H of all its
    coefficients (coeffs), and the reference's own benchmark method set_image + forward + coeffs
    (test/benchmark.py:141-162, test/bench.py:128-155), best of 3 on a one-image plan."""
    from pypwt_amd import Wavelets
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    x = synthetic_host_image(Nr, Nc)  # a 1D plan keeps its signal as one row: (1, N)
    w = Wavelets(x, wname, L, do_swt=swt, ndim=ndim)
    w.forward()
    _ = w.coeffs
    h2d = d2h = ref = None
    for _ in range(3):
        t0 = time.perf_counter(); w.set_image(x); t1 = time.perf_counter()
        w.forward(); w.synchronize(); t2 = time.perf_counter()
        _ = w.coeffs; t3 = time.perf_counter()
        h2d = min(h2d, t1 - t0) if h2d else t1 - t0
        d2h = min(d2h, t3 - t2) if d2h else t3 - t2
        ref = min(ref, t3 - t0) if ref else t3 - t0
    nbytes = x.nbytes
    coeff_planes = (3 * L + 1 if ndim == 2 else L + 1) if swt else 1  # DWT: as many coefficients as samples
    return {"h2d_ms": h2d * 1e3, "d2h_ms": d2h * 1e3, "h2d_GBps": nbytes / h2d / 1e9,
            "d2h_GBps": nbytes * coeff_planes / d2h / 1e9,
            "reference_method_ms": ref * 1e3,
            "reference_method_Msamples_s": Nr * Nc / ref / 1e6,
            "note": "pageable host memory; the reference times set_image + forward + coeffs (H2D + kernels + D2H)"}


def timed_steps(step, device_sync, steps):
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    return (time.perf_counter() - t0) / steps


def timed_region(step, device_sync, steps, dist, backend, before_closing_barrier=None, reheat=None):
    """The contract's timed region, the SAME code for one rank and for N: barrier + device sync, t0, exactly
    `steps` steps, device sync, t1, then the closing barrier + device sync.  Every rank's clock stops at t1,
    BEFORE the closing barrier: that barrier (an RCCL all-reduce kernel + a host sync, 50-100 us) only lines the
    ranks up again and is not part of anybody's interval, so the N = 1 and N > 1 figures are the same
    measurement.  Returns the max over ranks of t1 - t0 in seconds.

    `reheat`: a rank that reaches the opening barrier first idles its GPU until the last one arrives (the very first
    barrier also builds the RCCL communicator: hundreds of ms), and an MI355X that has idled for ~20 ms runs the next
    milliseconds 10 % slower (profiles/r03c_distgap.txt: 70.1 -> 77.4 us per step after a 20 ms sleep).  So the ranks
    are lined up TWICE: barrier, the same untimed re-heat on every rank (they stay together), then the barrier that
    opens the interval, which nobody waits in for more than its own latency."""
    if dist is not None and reheat is not None:
        barrier(dist, backend)
        device_sync()
        reheat()
    barrier(dist, backend)
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    elapsed = time.perf_counter() - t0
    if before_closing_barrier is not None:
        before_closing_barrier()
    barrier(dist, backend)
    device_sync()
    return max_over_ranks(elapsed, dist, backend)


def beyond_mall(cfg, device, steps, B=16):
    """The same step with B images per launch (16: 2.4 GiB of plan, far beyond the 256 MiB Infinity Cache): what the
    chip sustains from HBM rather than from its last-level cache.  B = 128 is BASELINE config 5's per-GPU shard -- the
    workload the multi-GPU runs of this script default to, measured here on one GPU so that the N-GPU lines have their
    own 1-GPU reference."""
    from pypwt_amd import BatchedWavelets
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    plan = BatchedWavelets(B, Nr, Nc, wname, L, do_swt=swt, ndim=ndim, device=device)
    plan.fill_hash(20240 + 2, 255.0)

    def step():
        plan.forward()
        if beta is not None:
            plan.soft_threshold(beta)
        plan.inverse()

    for _ in range(3):
        step()
    dt = timed_steps(step, plan.synchronize, max(5, min(steps, 20)))
    bytes_step = algorithmic_bytes_per_sample(cfg, threshold_separate=False) * B * Nr * Nc
    plan.cleanup()
    return {"batch": B, "ms_per_step": dt * 1e3, "us_per_image": dt / B * 1e6, "Msamples_s": B * Nr * Nc / dt / 1e6,
            "frac_of_hbm_peak": bytes_step / dt / 1e9 / HBM_PEAK_GBPS}


def other_config(name, device, steps):
    """One of the BASELINE configurations the line is NOT quoted on, measured in the same run on the same GPU: `steps` steps
    after a short pre-heat, the launch profile and the copy floor (kernel_profile).  A compact record for `extra.configs`."""
    from pypwt_amd import BatchedWavelets
    cfg = CONFIGS[name]
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    plan = BatchedWavelets(1, Nr, Nc, wname, L, do_swt=swt, ndim=ndim, device=device)
    try:
        plan.fill_hash(20240 + 2, 255.0)

        def step():
            plan.forward()
            if beta is not None:
                plan.soft_threshold(beta)
            plan.inverse()

        t0 = time.perf_counter()
        while (time.perf_counter() - t0) < 0.04:  # pre-heat
            for _ in range(20):
                step()
            plan.synchronize()
        regions = [timed_steps(step, plan.synchronize, steps) for _ in range(3)]
        step_s = median(regions)
        kernels, roofline, step_names = kernel_profile(plan, step, cfg, name, 1, steps, step_us=step_s * 1e6)
        thr_separate = any(n.startswith("soft_threshold") for n in step_names)
        bps = algorithmic_bytes_per_sample(cfg, threshold_separate=thr_separate)
        e2e_bytes = bps * Nr * Nc
        rec = {"workload": desc, "steps": steps, "ms_per_step": step_s * 1e3, "timed_regions_ms_per_step": [r * 1e3 for r in regions],
               "Msamples_s": Nr * Nc / step_s / 1e6, "algorithmic_bytes_per_sample": bps,
               "frac_of_hbm_peak": e2e_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
               "dominant_kernel": roofline.get("kernel"), "dominant_kernel_us": roofline.get("avg_us"),
               "dominant_kernel_us_basis": roofline.get("avg_us_basis"), "dominant_kernel_isolated_us": roofline.get("isolated_us"),
               "dominant_kernel_in_step_us": roofline.get("in_step_us"),
               "dominant_kernel_frac_of_hbm_peak": roofline.get("frac"), "copy_ceiling_GBps": roofline.get("copy_ceiling_GBps"),
               "copy_floor_us_per_step": roofline.get("step_copy_floor_us"),
               "kernels": [{"kernel": k["kernel"], "avg_us": k["avg_us"], "in_step_us": k.get("in_step_us"),
                            "isolated_us": k.get("isolated_us")} for k in kernels[:8]]}
        if roofline.get("step_copy_floor_us"):
            rec["copy_floor_over_step"] = roofline["step_copy_floor_us"] * 1e-6 / step_s
        return rec
    finally:
        plan.cleanup()


LONG_FILTER_PLANS = (("db20", 2048, 2048, 5), ("db20", 4096, 4096, 3))  # what the reference itself benchmarks: haar and db20 (test/benchmark.py:20-38)


def long_filters(device, steps=20):
    """The long-filter plans of VERDICT round 5, driver-timed: forward and inverse of db20 on 2048^2 (five levels) and
    4096^2 (three levels), pipelined, median of three regions of `steps` calls; which kernel family served every launch."""
    from pypwt_amd import BatchedWavelets
    recs = []
    for wname, Nr, Nc, L in LONG_FILTER_PLANS:
        plan = BatchedWavelets(1, Nr, Nc, wname, L, device=device)
        try:
            plan.fill_hash(20240 + 3, 255.0)

            def both():
                plan.forward()
                plan.inverse()
            for _ in range(5):
                both()
            plan.synchronize()
            fwd = median([timed_steps(plan.forward, plan.synchronize, steps) for _ in range(3)])
            fi = median([timed_steps(both, plan.synchronize, steps) for _ in range(3)])
            plan.enable_kernel_timing(True)
            plan.reset_kernel_times()
            both()
            fams = list(zip([n for n, _ in plan.kernel_times()], plan.kernel_families()))
            plan.enable_kernel_timing(False)
            bps = 16.0  # input read + coefficients written, coefficients read + image written
            recs.append({"workload": "%dx%d fp32 %s L%d" % (Nr, Nc, wname, plan.levels), "forward_us": fwd * 1e6,
                         "forward_inverse_us": fi * 1e6, "inverse_us": (fi - fwd) * 1e6,
                         "frac_of_hbm_peak": bps * Nr * Nc / fi / 1e9 / HBM_PEAK_GBPS,
                         "launches": ["%s:%s" % (n, f) if f else n for n, f in fams]})
        finally:
            plan.cleanup()
    return recs


SWT_FILTER_PLANS = (("db4", 2048, 2048, 4), ("sym8", 1080, 1920, 3), ("db20", 2048, 2048, 5))


def swt_filters(device, steps=20):
    """Round 6: the undecimated transform with filters of 6 taps and more, driver-timed -- the denoising step of the reference's
    documentation (doc/denoising.rst:85-141: forward, soft threshold, inverse) on db4 2048^2 L4 and sym8 1080 x 1920 L3, and the
    reference benchmark's own largest case (test/benchmark.py:24-38: swt2 db20 2048^2 at the maximum level, forward); pipelined,
    median of three regions of `steps` calls; the launch list says which kernels served every level.  Algorithmic bytes: 20 B per
    sample and level in each direction (one plane in, four out / four in, one out)."""
    from pypwt_amd import BatchedWavelets
    recs = []
    for wname, Nr, Nc, L in SWT_FILTER_PLANS:
        plan = BatchedWavelets(1, Nr, Nc, wname, L, do_swt=1, device=device)
        try:
            plan.fill_hash(20240 + 4, 255.0)

            def denoise():
                plan.forward()
                plan.soft_threshold(3.0)
                plan.inverse()
            for _ in range(5):
                denoise()
            plan.synchronize()
            fwd = median([timed_steps(plan.forward, plan.synchronize, steps) for _ in range(3)])
            step = median([timed_steps(denoise, plan.synchronize, steps) for _ in range(3)])
            plan.enable_kernel_timing(True)
            plan.reset_kernel_times()
            denoise()
            fams = list(zip([n for n, _ in plan.kernel_times()], plan.kernel_families()))
            plan.enable_kernel_timing(False)
            recs.append({"workload": "swt2 %dx%d fp32 %s L%d" % (Nr, Nc, wname, plan.levels), "forward_us": fwd * 1e6,
                         "forward_threshold_inverse_us": step * 1e6,
                         "forward_frac_of_hbm_peak": 20.0 * plan.levels * Nr * Nc / fwd / 1e9 / HBM_PEAK_GBPS,
                         "step_frac_of_hbm_peak": 40.0 * plan.levels * Nr * Nc / step / 1e9 / HBM_PEAK_GBPS,
                         "launches": ["%s:%s" % (n, f) if f else n for n, f in fams]})
        finally:
            plan.cleanup()
    return recs


def kernel_profile(plan, step, cfg, config_name, B, steps, step_us=None):
    """Per-launch shares of one step (HIP events on the plan's stream), the dominant kernel re-timed alone, its
    roofline and the copy ceiling measured beside it.  Returns (kernels, roofline, names of one step's launches)."""
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    # ---- per-kernel durations from HIP events on the plan's stream (second pass, same steps)
    plan.enable_kernel_timing(True)
    plan.reset_kernel_times()
    for _ in range(steps):
        step()
    times = plan.kernel_times(cap=64 * steps + 64)
    plan.enable_kernel_timing(False)
    plan.reset_kernel_times()
    per_step = len(times) // steps
    step_names = [n for n, _ in times[:per_step]]
    labels = None
    try:
        labels = labels_from_schedule(step_names, plan.schedule())
    except Exception:
        labels = None
    if labels is None:
        labels = label_step_kernels(step_names, L)
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
    event_overhead_us = in_step_durations(kernels, step_us)
    # The per-launch events above cost ~2.5 us of stream time each (the in-step durations sum to more than the step)
    # and the first launch behind an event absorbs its latency, so they RANK launches of similar length unreliably
    # (round 3 named the second-longest kernel).  The three longest candidates are therefore timed again on their
    # own -- `steps` launches of that level back to back between TWO HIP events on the plan's stream
    # (pdwt_time_level) -- and the longest of those is the dominant kernel.  That figure agrees with
    # rocprofv3 --kernel-trace (profiles/) and is the one the roofline uses.
    plan.forward()  # valid data in every buffer the levels read
    for k in kernels[:3]:
        lvl = level_of_kernel(k["kernel"], L)
        if lvl is not None:
            k["isolated_us"] = plan.time_level(lvl[0], inverse=lvl[1], reps=max(steps, 20))
    plan.inverse()
    dom = max(kernels[:3], key=lambda k: judged_duration(k)[0])
    dom_us, dom_basis = judged_duration(dom)
    dom_gbps = dom["algorithmic_bytes"] / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": dom_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": dom_gbps / HBM_PEAK_GBPS, "traffic": None, "kernel": dom["kernel"],
                "avg_us": dom_us, "avg_us_basis": dom_basis, "isolated_us": dom.get("isolated_us"), "in_step_us": dom.get("in_step_us"),
                "avg_us_in_step_with_event_overhead": dom["avg_us"], "event_overhead_us_per_launch": event_overhead_us,
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes"]}
    # The measured ceiling next to the spec peak: a plain 16-B grid-stride copy that moves the dominant launch's
    # bytes (half read, half written) out of this plan's own buffers, in this run, in the same cache state
    # (pdwt_time_copy).  The guide's figure for the kernel shape is 6.29 TB/s = 0.79 of the 8 TB/s peak.
    try:
        cap_elems = plan.copy_capacity()
        copy_elems = int(min(dom["algorithmic_bytes"] / 8, cap_elems))  # fp32: 8 bytes moved per value copied
        def copy_time(n):  # the faster of two timings: one stray interruption must not move the floor (seen: 35 us for 19)
            return min(plan.time_copy(n, reps=max(steps, 20)), plan.time_copy(n, reps=max(steps, 20)))
        copy_us = copy_time(copy_elems)
        copy_gbps = copy_elems * 8.0 / (copy_us * 1e-6) / 1e9
        roofline["copy_ceiling_GBps"] = copy_gbps
        roofline["copy_ceiling_us"] = copy_us
        roofline["copy_ceiling_bytes"] = copy_elems * 8.0
        roofline["frac_of_copy_ceiling"] = dom_gbps / copy_gbps if copy_gbps > 0 else None
        # ... and the same for EVERY launch of the step: what the step would take if each launch were a flat copy of its
        # own algorithmic bytes (small launches are bounded by the launch itself, which this includes)
        floor = 0.0
        for k in kernels:
            n = int(min(max(k["algorithmic_bytes"] / 8, 4), cap_elems))
            k["copy_us"] = copy_time(n) if k["algorithmic_bytes"] > 0 else 0.0
            # a launch that moves more than the plan's largest region (an SWT group of 11 planes): scaled by the byte ratio
            if n * 8.0 < k["algorithmic_bytes"]:
                k["copy_us"] *= k["algorithmic_bytes"] / (n * 8.0)
            floor += k["copy_us"]
        roofline["step_copy_floor_us"] = floor
    except Exception as e:  # an optional diagnostic must not break the line
        roofline["copy_ceiling_error"] = repr(e)
    # HBM bytes per launch of that kernel from the rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 + WRITE_SIZE, collected separately; tools/prof.sh + tools/summarize_pmc.py).
    # Counters cannot be read from inside the process, so the committed measurement is quoted.
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", "%s_traffic_%s.json" % (tag, config_name))
        if B == 1 and os.path.exists(tpath):
            try:
                # the profile labels launches by kernel family: "+soft" (a deferred threshold folded into the SWT
                # inverse) is a property of the plan state, not of the kernel
                t = json.load(open(tpath))["per_launch"].get(dom["kernel"].replace("+soft", ""))
                if t:
                    roofline["traffic"] = t["hbm_bytes"]
                    roofline["traffic_source"] = "profiles/" + os.path.basename(tpath)
                    break
            except Exception:
                pass
    return kernels, roofline, step_names


def dry_run(args, rank, world, dist, backend, out_stream=None):
    """No GPU: sleep-based steps so the launch / barrier / max-over-ranks / JSON path is testable
    with gloo on CPU."""
    cfg = CONFIGS[args.config]
    B, first, total = rank_batch(args, world, rank)
    step = lambda: time.sleep(0.001 * (1 + rank))
    alone_s = None
    if dist is not None:  # the same-workload one-rank reference, as in the real run
        barrier(dist, backend)
        if rank == 0:
            alone_s = median([timed_steps(step, lambda: None, args.steps) for _ in range(3)])
        barrier(dist, backend)
    regions = [timed_region(step, lambda: None, args.steps, dist, backend,
                            before_closing_barrier=lambda: time.sleep(1e-3 * rank * args.dry_run_barrier_skew_ms)) for _ in range(3)]
    dt = median(regions)
    if rank == 0:
        samples = total * cfg[0] * cfg[1]
        ref = None
        if alone_s is not None:
            one = B * cfg[0] * cfg[1] / alone_s / 1e6
            ref = {"one_gpu_same_workload_Msamples_s": one, "one_gpu_same_workload_ms_per_step": alone_s * 1e3,
                   "efficiency": samples / (dt / args.steps) / 1e6 / (world * one)}
        print(file=out_stream or sys.stdout, flush=True, *[json.dumps({"metric": "dry_run", "scaling_reference": ref, "value": samples / (dt / args.steps) / 1e6, "unit": "Msamples/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
                          "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": "dry-run", "config": args.config, "shard_rank0": [first, first + B],
                                     "images_per_step": total, "timed_region_ms": dt * 1e3,
                                     "timed_regions_ms": [t * 1e3 for t in regions]}})])


def claim_stdout():
    """Rank 0 must print ONE JSON line.  Libraries write to file descriptor 1 behind Python's back (RCCL prints a
    version banner when the communicator is built, the HIP runtime an occasional warning): point fd 1 at stderr for
    the whole run and return a file object on the ORIGINAL stdout for the JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


def single_process_main(args, out_stream):
    """--single-process: the product's own multi-GPU entry point (pypwt_amd.ShardedBatch) under the same contract -- W
    warm-up steps, exactly K timed steps between two synchronisations of EVERY device, one JSON line.  No torch, no
    process group: the shards exchange nothing."""
    from pypwt_amd import ShardedBatch
    n = args.gpus
    resolve_config(args, n)
    cfg = CONFIGS[args.config]
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    b = args.batch if args.batch is not None else DEFAULT_BATCH.get(args.config, 1)
    total = b if args.scaling == "strong" else b * n
    shared_gpu = os.environ.get("PDWT_BENCH_SHARE_GPU") == "1"
    S = ShardedBatch(total, Nr, Nc, wname, L, devices=[0] * n if shared_gpu else list(range(n)), do_swt=swt, ndim=ndim)
    S.fill_hash(20240 + 2, 255.0)

    def step():
        S.forward()
        if beta is not None:
            S.soft_threshold(beta)
        S.inverse()

    for _ in range(args.warmup):
        step()
    cold_s = timed_steps(step, S.synchronize, args.steps)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(20):
            step()
        S.synchronize()
    for _ in range(args.warmup):
        step()
    step_s = timed_steps(step, S.synchronize, args.steps)
    samples = total * Nr * Nc
    p0 = S.plans[0]

    def step0():
        p0.forward()
        if beta is not None:
            p0.soft_threshold(beta)
        p0.inverse()

    kernels, roofline, step_names = kernel_profile(p0, step0, cfg, args.config, p0.batch, args.steps)
    thr_separate = any(nm.startswith("soft_threshold") for nm in step_names)
    bps = algorithmic_bytes_per_sample(cfg, threshold_separate=thr_separate)
    per_gpu_bytes = bps * max(hi - lo for _, lo, hi in S.shards) * Nr * Nc
    out = {
        "metric": metric_name(args.config, desc, b),
        "value": samples / step_s / 1e6, "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (on-device index hash, 0..255)",
        "config": {"workload": "%s: %s" % (args.config, desc), "images_per_step": total,
                   "shards": [[d, lo, hi] for d, lo, hi in S.shards], "wavelet": wname, "levels": L, "shape": [Nr, Nc],
                   "parallelism": "image-sharded x%d, ONE process (pypwt_amd.ShardedBatch: one plan + stream + host thread "
                                  "per GPU), no collectives" % n,
                   "preheat_ms": args.preheat_ms, "timed_region_ms": step_s * args.steps * 1e3},
        "roofline": roofline,
        "end_to_end": {"algorithmic_bytes_per_sample": bps, "GBps_per_gpu": per_gpu_bytes / step_s / 1e9,
                       "frac_of_hbm_peak": per_gpu_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
                       "cold_ms_per_step": cold_s * 1e3},
        "kernels": kernels[:12],
    }
    if shared_gpu:
        out["config"]["shared_gpu_test_run"] = "all shards ran on GPU 0 (PDWT_BENCH_SHARE_GPU=1): not a multi-GPU measurement"
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    print(json.dumps(out), file=out_stream, flush=True)
    S.cleanup()
    return 0


def main():
    args = parse_args()
    if args.single_process and not args.dry_run:
        sys.exit(single_process_main(args, claim_stdout()))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))  # before any GPU / torch activity in this process
    out_stream = claim_stdout()
    rank, world, local_rank, dist, backend = init_dist(args)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; running %d rank(s) and reporting n_gpus = %d"
              % (args.gpus, world, world, world), file=sys.stderr)
    resolve_config(args, world)
    if args.dry_run:
        dry_run(args, rank, world, dist, backend, out_stream)
        if dist is not None:
            dist.destroy_process_group()
        return

    from pypwt_amd import BatchedWavelets
    cfg = CONFIGS[args.config]
    Nr, Nc, wname, L, swt, ndim, beta, desc = cfg
    B, first_image, total_images = rank_batch(args, world, rank)
    if B < 1:
        raise SystemExit("bench.py: rank %d has no image (--scaling strong with --batch %s over %d ranks)"
                         % (rank, args.batch, world))

    # PDWT_BENCH_SHARE_GPU=1 (tests on a one-GPU box, gloo backend): every rank on GPU 0 -- exercises the rank start-up, the
    # barriers and the aggregation on real hardware; the figure it prints is NOT a multi-GPU measurement and says so
    shared_gpu = world > 1 and os.environ.get("PDWT_BENCH_SHARE_GPU") == "1"
    plan = BatchedWavelets(B, Nr, Nc, wname, L, do_swt=swt, ndim=ndim, device=0 if shared_gpu else local_rank)
    # deterministic synthetic input generated ON the device; every rank gets different images
    plan.fill_hash(20240 + 2, 255.0, index_offset=first_image * Nr * Nc)

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

    barrier(dist, backend)  # builds the RCCL communicator now, not inside the measurement
    # cold figure: W warm-up steps after an idle period, then K steps (no pre-heat): what a one-shot caller sees
    for _ in range(args.warmup):
        step()
    cold_s = timed_steps(step, device_sync, args.steps)

    # pre-heat (untimed): sustained-throughput conditions for the timed steps
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(20):
            step()
        device_sync()
    for _ in range(args.warmup):
        step()
    device_sync()

    def reheat():  # ~50 ms of untimed steps, the same on every rank
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e3 < min(50.0, args.preheat_ms):
            for _ in range(20):
                step()
            device_sync()

    # Same-workload one-GPU reference for a multi-rank run: rank 0 times ITS shard alone (the other ranks wait at the barrier,
    # their GPUs idle), so that the line carries the denominator of its own scaling efficiency -- value(N) / value(1) of two
    # different default workloads (one image on one GPU, 128 per GPU on several) says nothing.
    alone_s = None
    if dist is not None:
        barrier(dist, backend)
        if rank == 0:
            reheat()
            alone_s = median([timed_steps(step, device_sync, args.steps) for _ in range(3)])
        barrier(dist, backend)
        reheat()

    # THREE timed regions, always, each the contract's (barrier, K steps, barrier): the line's figure is their median and all
    # three are in the line (a one-sided "time again when slow" would bias the figure downwards).
    timed_regions = [timed_region(step, device_sync, args.steps, dist, backend, reheat=reheat) for _ in range(3)]
    dt = median(timed_regions)
    cold_s = max_over_ranks(cold_s, dist, backend)

    step_s = dt / args.steps
    samples_per_step = total_images * Nr * Nc
    value = samples_per_step / step_s / 1e6

    kernels, roofline, step_names = kernel_profile(plan, step, cfg, args.config, B, args.steps, step_us=step_s * 1e6)
    # the soft threshold costs bytes only when it ran as its own sweep (a `soft_threshold` launch in the step)
    thr_separate = any(n.startswith("soft_threshold") for n in step_names)
    bps = algorithmic_bytes_per_sample(cfg, threshold_separate=thr_separate)
    e2e_bytes = bps * B * Nr * Nc  # per GPU per step
    e2e = {"algorithmic_bytes_per_sample": bps, "algorithmic_bytes_per_step_per_gpu": e2e_bytes,
           "GBps_per_gpu": e2e_bytes / step_s / 1e9,
           "frac_of_hbm_peak": e2e_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
           "cold_ms_per_step": cold_s * 1e3, "cold_Msamples_s": samples_per_step / cold_s / 1e6,
           "per_level_streaming_bytes_per_sample": per_level_streaming_bytes_per_sample(cfg, thr_separate),
           "sum_kernel_us": sum(k["avg_us"] for k in kernels)}
    if roofline.get("copy_ceiling_GBps"):
        e2e["frac_of_copy_ceiling"] = e2e["GBps_per_gpu"] / roofline["copy_ceiling_GBps"]
    if roofline.get("step_copy_floor_us"):
        # every launch of the step replaced by a flat copy of its algorithmic bytes, timed in this run on these buffers:
        # the step cannot be expected below this with one launch per level (group)
        e2e["copy_floor_us_per_step"] = roofline["step_copy_floor_us"]
        e2e["copy_floor_over_step"] = roofline["step_copy_floor_us"] * 1e-6 / step_s
    if swt and beta is not None:
        e2e["threshold"] = ("separate sweep" if thr_separate else
                            "folded into the inverse's loads (no launch, no bytes); as a separate sweep the step "
                            "would be charged %.0f B/sample" % algorithmic_bytes_per_sample(cfg, True))

    workload = args.config if args.config != "cfg5" else "cfg5 (per-GPU shard of cfg2 images)"
    out = {
        "metric": metric_name(args.config, desc, B),
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (on-device index hash, 0..255)",
        "config": {"workload": "%s: %s" % (workload, desc), "batch_per_gpu": B, "images_per_step": total_images,
                   "wavelet": wname, "levels": L, "shape": [Nr, Nc],
                   "parallelism": "image-sharded x%d, no collectives" % world, "preheat_ms": args.preheat_ms,
                   "timed_region_ms": dt * 1e3, "timed_regions_ms": [t * 1e3 for t in timed_regions],
                   "timed_region_rule": "median of three regions of `steps` steps each"},

        "roofline": roofline, "end_to_end": e2e, "kernels": kernels[:12],
    }
    if args.config in ("cfg2", "cfg5"):  # the configurations the north star is stated on
        out["target"] = target_record(e2e["frac_of_hbm_peak"], e2e_bytes, roofline.get("step_copy_floor_us"))
    if alone_s is not None:
        one = B * Nr * Nc / alone_s / 1e6
        out["scaling_reference"] = {"one_gpu_same_workload_Msamples_s": one, "one_gpu_same_workload_ms_per_step": alone_s * 1e3,
                                    "efficiency": value / (world * one),
                                    "note": "rank 0 alone on its shard of %d images per step (the other ranks waiting at a barrier), median "
                                            "of three regions; efficiency = value / (n_gpus x that)" % B}
    if dist is not None:
        if shared_gpu:
            out["config"]["shared_gpu_test_run"] = "all ranks ran on GPU 0 (PDWT_BENCH_SHARE_GPU=1): not a multi-GPU measurement"
        out["config"]["multi_gpu_note"] = ("one process per GPU; barrier (%s) before t0 and after t1, value = all ranks' "
                                           "samples / max over ranks of (t1 - t0)" % backend)
        if dt * 1e3 < 20.0:
            out["config"]["timed_region_warning"] = (
                "the timed region is %.2f ms: rank-to-rank start skew after the barrier (tens of us) is a visible "
                "share of it; use --steps >= %d or --config cfg5 (128 images per GPU and step) for a scaling figure"
                % (dt * 1e3, int(20.0 / (step_s * 1e3)) + 1))
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0 and world == 1 and not args.no_extras:
        extra = {}
        if not args.no_cpu_baseline:
            extra["cpu_baseline_1thread"] = cpu_baseline(cfg, threads=1)
            extra["pywt"] = pywt_baseline(args.config, cfg)
        try:
            extra["transfers"] = transfers(args.config, cfg, local_rank)
        except Exception as e:
            extra["transfers"] = {"error": repr(e)}
        if args.config == "cfg2" and B == 1:
            extra["configs"] = {}
            for other in ("cfg1", "cfg3", "cfg4"):  # the BASELINE configurations the line is not quoted on, same run, same GPU
                try:
                    extra["configs"][other] = other_config(other, local_rank, 20)
                except Exception as e:
                    extra["configs"][other] = {"error": repr(e)}
            try:
                extra["long_filters"] = long_filters(local_rank)
            except Exception as e:
                extra["long_filters"] = {"error": repr(e)}
            try:
                extra["swt_filters"] = swt_filters(local_rank)
            except Exception as e:
                extra["swt_filters"] = {"error": repr(e)}
            try:
                extra["beyond_infinity_cache"] = beyond_mall(cfg, local_rank, args.steps)
            except Exception as e:
                extra["beyond_infinity_cache"] = {"error": repr(e)}
            try:  # the multi-GPU default workload (cfg5: 128 images per GPU and step) on this one GPU
                extra["cfg5_shard_one_gpu"] = beyond_mall(cfg, local_rank, args.steps, B=128)
            except Exception as e:
                extra["cfg5_shard_one_gpu"] = {"error": repr(e)}
        out["extra"] = extra
    if rank == 0:
        print(json.dumps(out), file=out_stream, flush=True)
    plan.cleanup()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
