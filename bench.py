#!/usr/bin/env python3
"""Headline benchmark: room scene, Forward MIS, 1920x1080 (BASELINE.json configs[1]).

A "step" renders `--spp-per-step` samples per pixel of the full frame through the HIP wavefront
path (generate -> {extend, sort, shade<lobe>, shadow} x 10 -> accumulate), in batches of up to 2^27 paths
(16 spp); the default 16 steps x 64 spp are the whole 1024-spp job. All inputs (scene, BVH, queues) are resident in HBM before the
timed region starts; the film stays on the device (read-back is outside the timed region for
N = 1, the RCCL film reduce is inside it for N > 1).

N > 1 (launched by torch.distributed.run, one process per GPU): sample-pass sharding -- every rank
renders the full frame for its own sample indices (weak scaling: per-GPU work is fixed), then the
per-rank films are summed with one RCCL reduce over xGMI, the analogue of the per-worker film
sum of renderer::export_image (renderer.hpp:129-133).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PIPELINE_BYTES_PER_RAY = 112.0  # 32 ray in + 16 hit out + 64 path-state share
EXTEND_BYTES_PER_RAY = 48.0     # the extend kernel's own share of it (ray record in, hit record out)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 (vector)
# Fabric traffic of k_extend per ray, from the rocprofv3 PMC passes kept under profiles/
# (r01_pmc_v14_tcc_*.txt: TCC_EA0_RDREQ all 128-byte requests, TCC_EA0_WRREQ split 64/32 bytes; this is
# what FETCH_SIZE doubled + WRITE_SIZE give on gfx950): (1.12e8*128 + 2.07e8*64 + 1.60e8*32) B over
# 3.52e8 rays (123 B/ray before the queue columns went to streaming accesses). Counted at the L2's memory side, i.e. Infinity-Cache hits included.
EXTEND_TRAFFIC_BYTES_PER_RAY = 87.0


def cpu_baseline(S, A, args):
    """The reference (oracle/_ref, kind "reference") or the CPU restatement (kind "port") timed on
    the host cores on a bounded sample of the same workload: a centred crop of the same frame."""
    from oracle import pyref

    scene = S.room(args.width, args.height, tess=args.tess)
    kind = "reference" if pyref.have_reference() else "port"
    backend = pyref.reference() if kind == "reference" else pyref.restatement("f64")
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    sc = backend.scene(scene)
    tw, th = args.width // 4, args.height // 4
    tile = ((args.width - tw) // 2, (args.height - th) // 2, tw, th)
    spp = args.cpu_spp
    rgb, n, st = sc.render(spp, integrator=A.INTEGRATOR_MIS, max_path_length=10, seed=0, tile=tile, threads=cores)
    rays = st.closest_rays + st.any_rays
    out = {
        "value": rays / st.seconds / 1e6, "unit": "Mrays/s", "samples_per_s": n / st.seconds, "cores": cores, "kind": kind,
        "seconds": st.seconds, "rays_per_sample": rays / n,
        "sample": f"centre crop {tw}x{th} of the {args.width}x{args.height} frame, {spp} spp, Forward MIS, max_path_length 10, same scene and sample streams",
    }
    return out, rgb, tile, int(st.nan_samples)


def parity_gate(ctx, A, args, ref_rgb, tile, ref_nan, kind):
    """The correctness gate beside the speed: the GPU renders the CPU baseline's crop with the same
    sample indices and streams (untimed); per-pixel radiance is compared with the CPU result."""
    import numpy as np

    ctx.clear_film()
    ctx.reset_counters()
    ctx.render_pass(0, args.cpu_spp, A.INTEGRATOR_MIS, 10, True, tile)
    rgb, _ = ctx.read_film()
    c = ctx.counters()
    x0, y0, w, h = tile
    a, b = rgb[y0:y0 + h, x0:x0 + w], ref_rgb[y0:y0 + h, x0:x0 + w]
    floor = 1e-3 * max(float(np.abs(b).mean()), 1e-30)
    rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), floor)
    mse = float(np.mean((a - b) ** 2) / max(float(np.mean(b)) ** 2, 1e-30))
    return {
        "against": kind, "pixels": int(w * h), "spp": args.cpu_spp, "dtype": args.precision,
        "median_rel_err": float(np.median(rel)), "fraction_within_1e-2": float((rel <= 1e-2).mean()),
        "mean_ratio": float(a.mean() / b.mean()), "rel_mse": mse,
        "nan_samples": [int(c.nan_samples), ref_nan], "stack_overflows": int(c.stack_overflows),
    }


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
    ap.add_argument("--cpu-spp", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event stage timing (roofline leg)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if "WORLD_SIZE" in os.environ:  # launched by torch.distributed.run (also exercised with one rank)
        import torch
        import torch.distributed as dist

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
    if dist is not None:
        D = importlib.import_module("ray-tracing_amd.distributed")
        film = torch.as_tensor(ctx.film_device_view(), device=torch.device("cuda", local))

    spp = args.spp_per_step
    # sample indices: rank r owns [r * total, (r + 1) * total)
    total_steps = args.warmup + args.steps
    base = rank * total_steps * spp

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        ctx.render_pass(base + w * spp, spp)
    ctx.sync()
    if dist is not None and args.warmup > 0:
        # warm the collective too: RCCL sets its reduce channels up on first use, which belongs to warm-up
        D.reduce_film(film, args.width * args.height * spp * args.warmup, dst=0)
        torch.cuda.synchronize()
    ctx.clear_film()
    ctx.reset_counters()
    if not args.no_profile:
        ctx.set_profiling(True)
        ctx.stage_times()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ctx.render_pass(base + (args.warmup + k) * spp, spp)
    if dist is not None:
        ctx.sync()
        D.reduce_film(film, args.width * args.height * spp * args.steps, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0

    c = ctx.counters()
    stages = ctx.stage_times() if not args.no_profile else {}
    ctx.set_profiling(False)
    samples = args.width * args.height * spp * args.steps
    rays = c.closest_rays + c.shadow_rays
    if dist is not None:
        t = torch.tensor([elapsed, float(samples), float(rays)], dtype=torch.float64, device=torch.device("cuda", local))
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, samples, rays = float(tmax[0]), float(t[1]), float(t[2])

    if rank == 0:
        out = {
            "metric": "Mrays/s (room scene, Forward MIS, 1080p)", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "samples_per_s": samples / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {
                "workload": f"room scene stand-in (walls {12 * args.tess * args.tess} triangles, tess={args.tess}), Forward MIS, "
                            f"{args.width}x{args.height}, max_path_length 10, {spp} spp per step x {args.steps} steps per GPU",
                "spp_total_per_gpu": spp * args.steps, "rays_per_sample": rays / samples,
                "parallelism": "single GPU" if world == 1 else f"sample-pass sharding over {world} GPUs + RCCL film reduce",
            },
            "pipeline_hbm_fraction": rays / elapsed * PIPELINE_BYTES_PER_RAY / (HBM_PEAK_GBS * 1e9),
            "nan_samples": int(c.nan_samples), "stack_overflows": int(c.stack_overflows),
        }
        if stages:
            ms, launches = stages["extend"]
            rays_per_launch = c.closest_rays / max(1, launches)
            avg_s = ms * 1e-3 / max(1, launches)
            achieved = EXTEND_BYTES_PER_RAY * rays_per_launch / avg_s / 1e9
            out["roofline"] = {
                "kernel": "k_extend<float>" if args.precision == "f32" else "k_extend<double>", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": EXTEND_TRAFFIC_BYTES_PER_RAY * rays_per_launch if args.precision == "f32" else None,
                "traffic_note": "bytes per launch = 87 B/ray (PMC, profiles/r01_pmc_v14_tcc_*.txt) x rays per launch; the kernel is "
                                "VALU/latency bound (VALU busy 85 %, TA busy 78 %, lane utilisation 0.57: profiles/r01_pmc_v14_sq.txt), not HBM bound",
                "avg_launch_us": avg_s * 1e6, "launches": int(launches), "rays_per_launch": rays_per_launch,
                "algorithmic_bytes_per_ray": EXTEND_BYTES_PER_RAY,
                "stage_ms": {k: v[0] for k, v in stages.items()},
            }
        # VALU side of the roofline. Model flops: 25 per child-box slab test (4 per wide-node step),
        # 60 per primitive test (both counted on the device), 300 per ray for its half of a bounce's shading.
        local_rays = max(1, int(c.closest_rays + c.shadow_rays))
        flops_per_ray = (100.0 * int(c.reserved[3]) + 60.0 * int(c.reserved[4])) / local_rays + 300.0
        out["valu"] = {
            "model_flops_per_ray": flops_per_ray, "node_steps_per_ray": int(c.reserved[3]) / local_rays,
            "primitive_tests_per_ray": int(c.reserved[4]) / local_rays,
            "achieved": rays / elapsed * flops_per_ray / 1e12, "peak": VALU_PEAK_TFLOPS * world, "unit": "TFLOP/s",
            "frac": rays / elapsed * flops_per_ray / 1e12 / (VALU_PEAK_TFLOPS * world),
            "pmc": "k_extend: VALU busy 85 % of cycles at lane utilisation 0.57; k_shade: VALU busy 41 %, TA busy 71 % (profiles/r01_pmc_v14_sq.txt)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_rgb, tile, ref_nan = cpu_baseline(S, A, args)
            out["parity"] = parity_gate(ctx, A, args, ref_rgb, tile, ref_nan, out["cpu_baseline"]["kind"])
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
