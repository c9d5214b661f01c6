import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import A, R, S
scene = S.primitives_zoo(48, 48)
with R.GpuContext(0) as ctx:
    ctx.set_options(precision="f32", origin_offset="along_ray", epsilon=1e-4)
    ctx.upload_scene(scene)
    ctx.set_sampler(7)
    ctx.render_pass(0, 4, A.INTEGRATOR_MIS, 4, True, (22, 24, 2, 2))
    rgb, n = ctx.read_film()
    print("FILM", rgb[24:26, 22:24].ravel())
