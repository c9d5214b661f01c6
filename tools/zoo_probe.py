import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import A, R, S, load_film, pixel_rel_err
name = sys.argv[1] if len(sys.argv) > 1 else "zoo"
z, scene = load_film(name)
ref = z["mis_rgb_sum"]
for prec, mode in (("f32", "normal"), ("f32", "along_ray"), ("f64", "along_ray")):
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision=prec, origin_offset=mode, epsilon=1e-4)
        ctx.upload_scene(scene)
        ctx.set_sampler(int(z["seed"]))
        ctx.render_pass(0, int(z["spp"]), A.INTEGRATOR_MIS, int(z["max_path_length"]), True)
        rgb, n = ctx.read_film()
    rel = pixel_rel_err(rgb, ref)
    print(prec, mode, "median", np.median(rel), "within 1e-2", (rel <= 1e-2).mean(), "mean ratio", rgb.mean() / ref.mean())
    bad = rel > 1e-2
    for y in range(0, bad.shape[0], 2):
        print("".join("#" if bad[y, x] else "." for x in range(bad.shape[1])))
