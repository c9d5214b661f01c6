#!/usr/bin/env python3
"""Headline benchmark: room scene, Forward MIS, 1920x1080 (BASELINE.json configs[1]).

A "step" renders `--spp-per-step` samples per pixel of the full frame through the HIP wavefront path
(generate -> {extend, sort, shade<lobe>, shadow} x 10 -> accumulate) in batches of up to 2^27 paths (64 spp of a
1080p frame); the default 16 steps x 64 spp are the whole 1024-spp job. All inputs (scene, BVH, queues) are resident
in HBM before the timed region starts; the film stays on the device (read-back is outside the timed region for N = 1,
the RCCL film reduce is inside it for N > 1).

`--gpus N` with N > 1 starts N ranks itself (python -m torch.distributed.run, one process per GPU, started BEFORE
anything in this process touches a GPU) and relays rank 0's JSON line; under an external torch.distributed.run
(WORLD_SIZE set) it is a rank. Two partitionings of the job, the multi-GPU analogue of the reference's N worker
threads whose films are summed at export (renderer.hpp:65-81, 125-135):
  --shard samples (default; weak scaling)  every rank renders the full frame for its own sample indices, the films
                                           are summed with one RCCL reduce over xGMI;
  --shard tiles   (strong scaling)         ONE job of steps x spp samples per pixel; rank r renders the bands of
                                           --band-rows rows with band % N == r (fcgpu_render_bands: full-width tiles
                                           dealt round-robin, one pass per step), the same reduce gathers the disjoint
                                           films, and rank 0 re-renders a few bands it does not own and checks them
                                           against the gathered film bit for bit.

Prints ONE JSON line on rank 0 and exits non-zero when the fp32 parity gate beside the speed fails.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PIPELINE_BYTES_PER_RAY = 112.0  # 32 ray in + 16 hit out + 64 path-state share
EXTEND_BYTES_PER_RAY = 48.0     # the extend kernel's own share of it (ray record in, hit record out)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 (vector)
# rocprofv3 PMC passes of THIS build at the bench's own batch size, reduced to per-kernel numbers by
# tools/pmc_summary.py; the passes themselves are kept under profiles/ (named in the file's "source")
PMC_SUMMARY = os.path.join(ROOT, "ray-tracing_amd", "pmc_summary.json")

# fp32 parity gate (same sample streams, against the CPU baseline's film). ONE bar, stated here, asserted with the same
# numbers by tests/test_gpu_headline.py on the committed reference crop and quoted in DESIGN.md section 4: at 16 spp at
# least 99.4 % of the pixels within 1e-2 and 98.7 % within 1e-3 of the reference (measured 99.7 % / 99.2 %), median relative
# error <= 2e-6 (measured 4e-7), image mean within 3e-4. A pixel is the sum of its paths and ONE path whose discrete
# decision flips on a rounding (a hit across a triangle edge: 1.9e-3 of the paths) moves it, so the allowed fractions
# beyond the bounds grow in proportion to the samples per pixel; they are capped, so that the gate can always fail.
PARITY_MEDIAN_REL_ERR = 2e-6
PARITY_WITHIN_1E2_MIN_16SPP = 0.994
PARITY_WITHIN_1E3_MIN_16SPP = 0.987
PARITY_MEAN_RATIO = 3e-4
PARITY_REL_MSE = 1e-3


def parity_fraction_floors(spp):
    """(within 1e-2 >=, within 1e-3 >=) at `spp` samples per pixel: the 16-spp bars scaled by the path count, never below 0.5."""
    k = spp / 16.0
    return max(0.5, 1.0 - (1.0 - PARITY_WITHIN_1E2_MIN_16SPP) * k), max(0.5, 1.0 - (1.0 - PARITY_WITHIN_1E3_MIN_16SPP) * k)


def visible_gpus():
    """GPUs a rank could use, counted in a throw-away child process: the launcher itself never loads a GPU runtime, so
    whatever counting devices initialises dies with that child."""
    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, text=True)
    try:
        return int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(args):
    """--gpus N without an external launcher: N fresh ranks via torch.distributed.run. Nothing in this process touches a
    GPU (devices are counted by a child), so no process that holds a GPU context is ever replaced."""
    have = visible_gpus()
    if have < args.gpus and not (args.backend == "gloo" and have >= 1):
        print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible: refusing to run fewer ranks than asked", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def cpu_model():
    """'model name (family F model M)' of the first CPU in /proc/cpuinfo: virtualised hosts shorten the name ("AMD EPYC"),
    the family / model numbers still say which part it is."""
    want = {"model name": None, "cpu family": None, "model": None}
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k = ln.split(":", 1)[0].strip()
                if k in want and want[k] is None:
                    want[k] = ln.split(":", 1)[1].strip()
                if all(v is not None for v in want.values()):
                    break
    except OSError:
        pass
    if want["model name"] is None:
        return "unknown"
    return f'{want["model name"]} (family {want["cpu family"]} model {want["model"]})'


def cpu_threads():
    """Worker threads of the CPU baseline = this job's share of the host, as SURVEY 8(d) asks (the reference's
    worker_count = hardware_concurrency()): the CPUs the process may run on, capped by OMP_NUM_THREADS where the
    operator set it (on the GPU box it is the one-GPU share of the machine, 16, while os.cpu_count() shows all of the
    host's logical CPUs; more threads than the share only time-slice). All three numbers go into the line."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or affinity
    return max(1, min(share, affinity)), logical, affinity


def cpu_baseline(S, A, args):
    """The reference (oracle/_ref, kind "reference") or the CPU restatement (kind "port") timed on the host cores on a
    bounded sample of the same workload: a centred crop of the same frame, same scene and sample streams. Two legs
    on all cores (the first one's film is what the parity gate compares) sized for >= 30 s together, and one leg on a
    single thread."""
    from oracle import pyref

    scene = S.room(args.width, args.height, tess=args.tess)
    kind = "reference" if pyref.have_reference() else "port"
    backend = pyref.reference() if kind == "reference" else pyref.restatement("f64")
    cores, logical, affinity = cpu_threads()
    sc = backend.scene(scene)
    tw, th = args.width // 4, args.height // 4
    tile = ((args.width - tw) // 2, (args.height - th) // 2, tw, th)
    legs = []
    rgb, n, st = sc.render(args.parity_spp, integrator=A.INTEGRATOR_MIS, max_path_length=10, seed=0, tile=tile, threads=cores)
    legs.append((n, st))
    if args.cpu_spp > args.parity_spp:
        _, n2, st2 = sc.render(args.cpu_spp - args.parity_spp, integrator=A.INTEGRATOR_MIS, max_path_length=10, seed=0, tile=tile,
                               threads=cores, sample_begin=args.parity_spp)
        legs.append((n2, st2))
    seconds = sum(s.seconds for _, s in legs)
    samples = sum(k for k, _ in legs)
    rays = sum(s.closest_rays + s.any_rays for _, s in legs)
    out = {
        "value": rays / seconds / 1e6, "unit": "Mrays/s", "samples_per_s": samples / seconds, "cores": cores, "kind": kind,
        "seconds": seconds, "rays_per_sample": rays / samples, "cpu_model": cpu_model(),
        "host_logical_cpus": logical, "affinity_cpus": affinity, "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
        "cores_rule": "min(OMP_NUM_THREADS = the job's share of the host, CPUs in the affinity mask)",
        "sample": f"centre crop {tw}x{th} of the {args.width}x{args.height} frame, {args.cpu_spp} spp, Forward MIS, max_path_length 10, "
                  "same scene and sample streams",
    }
    if args.cpu_1t_spp > 0:
        _, n1, st1 = sc.render(args.cpu_1t_spp, integrator=A.INTEGRATOR_MIS, max_path_length=10, seed=0, tile=tile, threads=1)
        r1 = st1.closest_rays + st1.any_rays
        out["one_thread"] = {"value": r1 / st1.seconds / 1e6, "unit": "Mrays/s", "samples_per_s": n1 / st1.seconds, "seconds": st1.seconds,
                             "sample": f"the same crop, {args.cpu_1t_spp} spp"}
    return out, rgb, tile, int(legs[0][1].nan_samples)


def parity_gate(ctx, A, args, ref_rgb, tile, ref_nan, kind):
    """The correctness gate beside the speed: the GPU renders the CPU baseline's crop with the same
    sample indices and streams (untimed); per-pixel radiance is compared with the CPU result against the thresholds
    stated at the top of this file."""
    import numpy as np

    ctx.clear_film()
    ctx.reset_counters()
    ctx.render_pass(0, args.parity_spp, A.INTEGRATOR_MIS, 10, True, tile)
    rgb, _ = ctx.read_film()
    c = ctx.counters()
    x0, y0, w, h = tile
    a, b = rgb[y0:y0 + h, x0:x0 + w], ref_rgb[y0:y0 + h, x0:x0 + w]
    floor = 1e-3 * max(float(np.abs(b).mean()), 1e-30)
    rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), floor)
    mse = float(np.mean((a - b) ** 2) / max(float(np.mean(b)) ** 2, 1e-30))
    out = {
        "against": kind, "pixels": int(w * h), "spp": args.parity_spp, "dtype": args.precision,
        "median_rel_err": float(np.median(rel)), "fraction_within_1e-2": float((rel <= 1e-2).mean()),
        "fraction_within_1e-3": float((rel <= 1e-3).mean()),
        "mean_ratio": float(a.mean() / b.mean()), "rel_mse": mse,
        "nan_samples": [int(c.nan_samples), ref_nan], "stack_overflows": int(c.stack_overflows), "medium_overflows": int(c.medium_overflows),
    }
    f64 = args.precision == "f64"
    floor2, floor3 = parity_fraction_floors(args.parity_spp)
    th = {
        "median_rel_err_max": 1e-12 if f64 else PARITY_MEDIAN_REL_ERR,
        "fraction_within_1e-2_min": 1.0 if f64 else floor2, "fraction_within_1e-3_min": 1.0 if f64 else floor3,
        "mean_ratio_tol": 1e-9 if f64 else PARITY_MEAN_RATIO, "rel_mse_max": 1e-18 if f64 else PARITY_REL_MSE,
    }
    out["thresholds"] = th
    out["pass"] = bool(out["median_rel_err"] <= th["median_rel_err_max"] and out["fraction_within_1e-2"] >= th["fraction_within_1e-2_min"]
                       and out["fraction_within_1e-3"] >= th["fraction_within_1e-3_min"]
                       and abs(out["mean_ratio"] - 1.0) <= th["mean_ratio_tol"] and mse <= th["rel_mse_max"]
                       and out["nan_samples"][0] == out["nan_samples"][1] and out["stack_overflows"] == 0 and out["medium_overflows"] == 0)
    return out


def f64_leg(R, S, A, args, device):
    """The reference-precision rate on the same workload (a short extra leg after the timed region): fp64 kernels,
    the reference's own arithmetic (matches it per pixel to 1e-9, tests/test_gpu_headline.py)."""
    spp = 32  # 2^26 paths: the fp64 queues of one 64-spp batch would be 57 GB
    ctx = R.GpuContext(device)
    ctx.set_options(precision="f64")
    ctx.upload_scene(S.room(args.width, args.height, tess=args.tess))
    ctx.set_sampler(0)
    ctx.render_pass(0, 4)
    ctx.sync()
    ctx.clear_film()
    ctx.reset_counters()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.render_pass(4, spp)
    ctx.sync()
    dt = time.perf_counter() - t0
    c = ctx.counters()
    ctx.close()
    n = args.width * args.height * spp
    return {"dtype": "f64", "spp": spp, "samples_per_s": n / dt, "value": (c.closest_rays + c.shadow_rays) / dt / 1e6, "unit": "Mrays/s",
            "nan_samples": int(c.nan_samples)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tess", type=int, default=64)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--shard", default="samples", choices=["samples", "tiles"])
    ap.add_argument("--band-rows", type=int, default=4, help="tile sharding: rows per band (full-width tiles dealt round-robin)")
    ap.add_argument("--cpu-spp", type=int, default=208, help="CPU baseline: samples per pixel of the crop on all cores (>= 30 s of work)")
    ap.add_argument("--cpu-1t-spp", type=int, default=4, help="CPU baseline: samples per pixel of the crop on one thread (0 = skip)")
    ap.add_argument("--parity-spp", type=int, default=16, help="samples per pixel the parity gate compares (first leg of the CPU baseline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one GPU per rank (the measured path). gloo = rehearsal of the N-rank logic on a box with fewer "
                         "GPUs than ranks: ranks share GPUs (local rank modulo the GPUs visible) and the films are reduced through the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-leg", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event stage timing (roofline leg)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    # stdout carries the ONE result line and nothing else: whatever libraries print there (RCCL's version banner at the
    # first communicator, for one) is sent to stderr; the line itself goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): the two must agree", file=sys.stderr)
        sys.exit(2)
    dist = None
    torch = None
    if "WORLD_SIZE" in os.environ:  # launched by torch.distributed.run (also exercised with one rank)
        import torch
        import torch.distributed as dist

        if args.backend == "gloo":
            local = local % max(1, torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() <= local:
                print(f"bench.py: rank {rank} has no GPU {local} ({torch.cuda.device_count()} visible)", file=sys.stderr)
                sys.exit(2)
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    S = importlib.import_module("ray-tracing_amd.scenes")
    A = importlib.import_module("ray-tracing_amd._abi")
    R = importlib.import_module("ray-tracing_amd.renderer")

    scene = S.room(args.width, args.height, tess=args.tess)
    ctx = R.GpuContext(local)
    ctx.set_options(precision=args.precision)
    ctx.upload_scene(scene)
    ctx.set_sampler(0)
    film = None
    nccl = dist is not None and args.backend == "nccl"
    if dist is not None:
        D = importlib.import_module("ray-tracing_amd.distributed")
    if nccl:
        film = torch.as_tensor(ctx.film_device_view(), device=torch.device("cuda", local))
    reduced_host = [None]

    def reduce_films():
        """per-rank films -> rank 0: in place on the device over RCCL, or (gloo rehearsal) through host copies"""
        if nccl:
            D.reduce_film(film, 0, dst=0)
        else:
            host = torch.from_numpy(ctx.read_film()[0])
            D.reduce_film(host, 0, dst=0)
            reduced_host[0] = host.numpy()

    spp = args.spp_per_step
    total_steps = args.warmup + args.steps
    tiles = args.shard == "tiles"
    # samples: rank r owns the sample indices [r * total, (r + 1) * total) of every pixel;
    # tiles:   every rank renders the same sample indices [0, total) of the pixels it owns
    base = 0 if tiles else rank * total_steps * spp

    def render_steps(first, count):
        """steps [first, first + count). samples: one pass of `spp` samples of the full frame per step. tiles: a rank owns
        1 / world of the pixels, so it takes the samples of `world` steps in ONE pass: its batches stay at the 2^27 paths
        they have on one GPU (a pass of spp samples over 1 / 8 of the frame would run 16.6 M-path batches at N = 8, which
        the batch-size table of DESIGN.md section 5 prices at -14 %); the film does not depend on how the sample range
        is cut into passes (one writer per pixel, samples added in index order)."""
        if not tiles:
            for k in range(first, first + count):
                ctx.render_pass(base + k * spp, spp)
            return
        k = first
        while k < first + count:
            n = min(world, first + count - k)
            ctx.render_bands(args.band_rows, rank, world, base + k * spp, n * spp)
            k += n

    def barrier():
        ctx.sync()
        if dist is not None:
            if nccl:
                torch.cuda.synchronize()
            dist.barrier()
            if nccl:
                torch.cuda.synchronize()

    render_steps(0, args.warmup)
    ctx.sync()
    if dist is not None and args.warmup > 0:
        # warm the collective too: RCCL sets its reduce channels up on first use, which belongs to warm-up
        reduce_films()
        if nccl:
            torch.cuda.synchronize()
    ctx.clear_film()
    ctx.reset_counters()
    if not args.no_profile:
        ctx.set_profiling(True)
        ctx.stage_times()

    barrier()
    t0 = time.perf_counter()
    render_steps(args.warmup, args.steps)
    reduce_s = 0.0
    if dist is not None:
        ctx.sync()
        t1 = time.perf_counter()
        reduce_films()
        if nccl:
            torch.cuda.synchronize()
        reduce_s = time.perf_counter() - t1
    barrier()
    elapsed = time.perf_counter() - t0

    c = ctx.counters()
    stages = ctx.stage_times() if not args.no_profile else {}
    ctx.set_profiling(False)
    local_samples = int(c.samples)
    rays = c.closest_rays + c.shadow_rays
    samples = local_samples
    if dist is not None:
        t = torch.tensor([elapsed, float(samples), float(rays), reduce_s], dtype=torch.float64, device=torch.device("cuda", local) if nccl else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, samples, rays, reduce_s = float(tmax[0]), float(t[1]), float(t[2]), float(tmax[3])

    tile_check = None
    if tiles and rank == 0:
        # the gathered film against a rank-local re-render of bands another rank owns (for N = 1: against the same rows
        # rendered as a plain tile): the per-pixel sums must be bit-identical
        import numpy as np

        ctx.sync()
        gathered = ctx.read_film()[0] if reduced_host[0] is None else reduced_host[0]
        nbands = (args.height + args.band_rows - 1) // args.band_rows
        picks = sorted({b for b in (1 % nbands, nbands // 2 | 1, nbands - 1) if b < nbands and (world == 1 or b % world != 0)})
        ctx.clear_film()
        for b in picks:
            y0 = b * args.band_rows
            for k in range(args.steps):
                ctx.render_pass(base + (args.warmup + k) * spp, spp, tile=(0, y0, args.width, min(args.band_rows, args.height - y0)))
        local_film, _ = ctx.read_film()
        rows = np.concatenate([np.arange(b * args.band_rows, min((b + 1) * args.band_rows, args.height)) for b in picks])
        tile_check = {"bands_checked": picks, "owners": [b % world for b in picks], "rows": int(len(rows)),
                      "bit_identical": bool(np.array_equal(gathered[rows], local_film[rows])),
                      "film_mean": float(gathered.mean())}

    rc = 0
    if rank == 0:
        expected = args.width * args.height * spp * args.steps * (1 if tiles else world)
        out = {
            "metric": "Mrays/s (room scene, Forward MIS, 1080p)", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "samples_per_s": samples / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {
                "workload": f"room scene stand-in (walls {12 * args.tess * args.tess} triangles, tess={args.tess}), Forward MIS, "
                            f"{args.width}x{args.height}, max_path_length 10, {spp} spp per step x {args.steps} steps"
                            + (" (one job shared by all GPUs)" if tiles else " per GPU"),
                "spp_total_per_gpu": spp * args.steps, "rays_per_sample": rays / samples, "shard": args.shard,
                "parallelism": "single GPU" if world == 1 else (
                    f"tile sharding: bands of {args.band_rows} rows dealt round-robin over {world} ranks + film gather" if tiles else
                    f"sample-pass sharding over {world} ranks + film reduce") + (" (RCCL over xGMI, one GPU per rank)" if nccl else
                                                                                  " (gloo rehearsal: ranks share GPUs, films reduced through the host)"),
                "rccl_ranks": dist.get_world_size() if nccl else 0, "backend": args.backend if dist is not None else None,
                # paths of one rank's largest pass (the library cuts it into batches of at most 2^27)
                "paths_per_pass_per_rank": (args.width * ((args.height + world - 1) // world) * spp * min(world, args.steps)) if tiles
                                           else args.width * args.height * spp,
            },
            "reduce_ms": 1e3 * reduce_s if dist is not None else None,
            "film_bytes": 24 * args.width * args.height,
            "samples_rendered": samples, "samples_expected": expected,
            "pipeline_hbm_fraction": rays / elapsed * PIPELINE_BYTES_PER_RAY / (HBM_PEAK_GBS * 1e9 * world),
            "nan_samples": int(c.nan_samples), "stack_overflows": int(c.stack_overflows), "medium_overflows": int(c.medium_overflows),
        }
        if c.stack_overflows or c.medium_overflows:
            print("bench.py: dropped traversal-stack or medium-stack pushes: the frames are not valid", file=sys.stderr)
            rc = 1
        if samples != expected:
            print(f"bench.py: rendered {samples} samples, expected {expected}", file=sys.stderr)
            rc = 1
        if tile_check is not None:
            out["tile_check"] = tile_check
            if not tile_check["bit_identical"]:
                print("bench.py: the gathered film differs from the rank-local re-render", file=sys.stderr)
                rc = 1
        # counters of the rocprofv3 PMC passes belong beside this run's timing only if they were taken on THIS build
        pmc = None
        build = importlib.import_module("ray-tracing_amd.buildinfo").kernel_source_hash()
        pmc_stale = None
        if os.path.exists(PMC_SUMMARY):
            with open(PMC_SUMMARY) as f:
                pmc = json.load(f)
            pmc_stale = pmc.get("build") != build
            if pmc_stale:
                print(f"bench.py: {PMC_SUMMARY} describes build {pmc.get('build')}, this tree is {build}: its counters are not printed", file=sys.stderr)
                pmc = {"source": pmc.get("source"), "build": pmc.get("build")}
        out["build"] = build
        out["pmc_stale"] = pmc_stale
        if stages:
            ms, launches = stages["extend"]
            rays_per_launch = c.closest_rays / max(1, launches)
            avg_s = ms * 1e-3 / max(1, launches)
            achieved = EXTEND_BYTES_PER_RAY * rays_per_launch / avg_s / 1e9
            k = (pmc or {}).get("kernels", {}).get("k_extend", {}) if args.precision == "f32" else {}
            out["roofline"] = {
                "kernel": "k_extend<float>" if args.precision == "f32" else "k_extend<double>", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                # fabric bytes per launch from the PMC passes (TCC_EA0_RDREQ x 128 B + WRREQ 64 / 32 B), per ray x this run's rays per launch
                "traffic": k["fabric_bytes_per_ray"] * rays_per_launch if k.get("fabric_bytes_per_ray") else None,
                "pmc_source": (pmc or {}).get("source"), "pmc_build": (pmc or {}).get("build"), "pmc_stale": pmc_stale,
                "pmc_batch_paths": (pmc or {}).get("batch_paths"),
                "avg_launch_us": avg_s * 1e6, "launches": int(launches), "rays_per_launch": rays_per_launch,
                "algorithmic_bytes_per_ray": EXTEND_BYTES_PER_RAY,
                "stage_ms": {k2: v[0] for k2, v in stages.items()},
            }
            # The same kernel against the roof that does bound it (VALU issue under divergence, beside the L1
            # address path): model flops = 100 per 4-wide node step + 60 per primitive test, both counted on the device,
            # over the same HIP-event launch time; beside it what the PMC passes of this build measured.
            local_rays = max(1, int(c.closest_rays + c.shadow_rays))
            trav_flops_per_ray = (100.0 * int(c.reserved[3]) + 60.0 * int(c.reserved[4])) / local_rays
            achieved_tf = trav_flops_per_ray * rays_per_launch / avg_s / 1e12
            out["roofline_valu"] = {
                "kernel": out["roofline"]["kernel"], "bound": "valu", "achieved": achieved_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / VALU_PEAK_TFLOPS, "model_flops_per_ray": trav_flops_per_ray,
                "valu_busy": k.get("valu_busy"), "lane_util": k.get("lane_util"), "ta_busy": k.get("ta_busy"),
                "useful_issue_frac": (k["valu_busy"] * k["lane_util"]) if k.get("valu_busy") and k.get("lane_util") else None,
                "pmc_stale": pmc_stale,
            }
        # VALU side of the whole pipeline. Model flops: 25 per child-box slab test (4 per wide-node step),
        # 60 per primitive test (both counted on the device), 300 per ray for its half of a bounce's shading.
        local_rays = max(1, int(c.closest_rays + c.shadow_rays))
        flops_per_ray = (100.0 * int(c.reserved[3]) + 60.0 * int(c.reserved[4])) / local_rays + 300.0
        out["valu"] = {
            "model_flops_per_ray": flops_per_ray, "node_steps_per_ray": int(c.reserved[3]) / local_rays,
            "primitive_tests_per_ray": int(c.reserved[4]) / local_rays,
            "achieved": rays / elapsed * flops_per_ray / 1e12, "peak": VALU_PEAK_TFLOPS * world, "unit": "TFLOP/s",
            "frac": rays / elapsed * flops_per_ray / 1e12 / (VALU_PEAK_TFLOPS * world),
            "pmc": {name: {f: v.get(f) for f in ("valu_busy", "lane_util", "ta_busy", "wait_frac", "fabric_bytes_per_ray", "avg_us")}
                    for name, v in (pmc or {}).get("kernels", {}).items()} if pmc and pmc.get("kernels") else None,
            "pmc_source": (pmc or {}).get("source"), "pmc_stale": pmc_stale,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_rgb, tile, ref_nan = cpu_baseline(S, A, args)
            out["parity"] = parity_gate(ctx, A, args, ref_rgb, tile, ref_nan, out["cpu_baseline"]["kind"])
            if not out["parity"]["pass"]:
                print("bench.py: the parity gate FAILED: " + json.dumps(out["parity"]), file=sys.stderr)
                rc = 1
    ctx.close()
    if rank == 0:
        if world == 1 and not args.no_f64_leg and args.precision == "f32":
            out["f64"] = f64_leg(R, S, A, args, local)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
