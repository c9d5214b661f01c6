import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import A, R, S
scene = S.primitives_zoo(48, 48)
for mpl in (2, 3, 4, 5, 6, 10):
    out = {}
    for prec in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=prec, origin_offset="along_ray", epsilon=1e-4)
            ctx.upload_scene(scene)
            ctx.set_sampler(7)
            ctx.render_pass(0, 16, A.INTEGRATOR_MIS, mpl, True)
            rgb, n = ctx.read_film()
            c = ctx.counters()
        out[prec] = rgb[16:34, 16:30]
        print(mpl, prec, "rays", c.closest_rays, c.shadow_rays, c.shadow_occluded, "nan", c.nan_samples)
    print("max_path_length", mpl, "blob mean ratio f32/f64", out["f32"].mean() / out["f64"].mean())
