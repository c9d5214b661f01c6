"""One-off differential fuzzing (development tool): GPU fp64 and fp32 against the CPU restatement on many random scenes.
usage: python tools/fuzz_random_scenes.py FIRST_SEED COUNT"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
S = importlib.import_module("ray-tracing_amd.scenes")
A = importlib.import_module("ray-tracing_amd._abi")
R = importlib.import_module("ray-tracing_amd.renderer")
from oracle import pyref
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    sc = S.random_scene(seed, 32, 24)
    o = pyref.restatement("f64").scene(sc)
    ref, n, st = o.render(3, max_path_length=8, seed=seed, threads=8)
    out = []
    for prec in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            # every third seed on the device-built hierarchy, every fourth on the 8-wide node
            ctx.set_options(precision=prec, bvh_builder="device" if seed % 3 == 0 else "host", bvh_width=8 if seed % 4 == 1 else 4)
            ctx.upload_scene(sc)
            ctx.set_sampler(seed)
            ctx.render_pass(0, 3, A.INTEGRATOR_MIS, 8, True)
            rgb, m = ctx.read_film()
            c = ctx.counters()
        rel = np.abs(rgb - ref).max(2) / np.maximum(np.abs(ref).max(2), 1e-3 * ref.mean())
        out.append((prec, float(rel.max()), float(np.median(rel)), float((rel <= 1e-2).mean()), int(c.closest_rays) - int(st.closest_rays), int(c.nan_samples) - int(st.nan_samples), int(c.stack_overflows) + int(c.medium_overflows)))
    ok = out[0][1] <= 1e-9 and out[0][4] == 0 and out[0][5] == 0 and out[0][6] == 0 and out[1][3] >= 0.85 and out[1][6] == 0
    if not ok and out[0][1] <= 1e-6 and out[0][4:] == (0, 0, 0) and out[1][3] >= 0.85 and out[1][6] == 0:
        # only the fp64 bar missed, by little, with equal ray counts: wrong, or an ill-conditioned pixel? The restatement
        # under +-1 ulp noise on its transcendentals (the distance between the device's libm and glibc) answers
        pr = pyref.restatement("f64")
        spread = 0.0
        for noise in range(1, 9):
            pr.set_ulp_noise(noise)
            try:
                alt, _, _ = o.render(3, max_path_length=8, seed=seed, threads=8)
            finally:
                pr.set_ulp_noise(0)
            spread = max(spread, float((np.abs(alt - ref).max(2) / np.maximum(np.abs(ref).max(2), 1e-3 * ref.mean())).max()))
        print(seed, "fp64 worst pixel %.2e; the restatement moves by up to %.2e under 1-ulp libm noise" % (out[0][1], spread), flush=True)
        ok = out[0][1] <= 4.0 * spread
    bad += 0 if ok else 1
    if not ok or seed % 10 == 0:
        print(seed, "OK" if ok else "MISMATCH", out, flush=True)
print("done:", count, "scenes,", bad, "mismatching", flush=True)
