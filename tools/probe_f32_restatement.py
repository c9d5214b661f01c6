import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from common import *
from oracle import pyref
scene = S.room(width=64, height=36, tess=4)
for along in (True, False):
    o = pyref.restatement("f32").scene(scene, along_ray=along)
    ref, n, _ = o.render(4, seed=2, threads=4)
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32", origin_offset="along_ray" if along else "normal")
        ctx.upload_scene(scene); ctx.set_sampler(2); ctx.render_pass(0, 4)
        rgb, m = ctx.read_film()
    rel = pixel_rel_err(rgb, ref)
    print("f32 device vs f32 restatement, along_ray", along, "median", np.median(rel), "within 1e-2", (rel <= 1e-2).mean(), "within 1e-4", (rel<=1e-4).mean(), "within 1e-5", (rel<=1e-5).mean())
