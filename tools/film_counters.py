import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_gpu_parity import load_film, make_ctx, film_stratified, A
for name in ("normals_diffuse", "normals_mirror"):
    z, scene = load_film(name)
    with make_ctx(scene, "f64") as ctx:
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        for tag, integ in (("mis", A.INTEGRATOR_MIS), ("bsdf", A.INTEGRATOR_BSDF)):
            ctx.clear_film(); ctx.reset_counters()
            ctx.render_pass(0, int(z["spp"]), integ, int(z["max_path_length"]), True)
            c = ctx.counters(); r = z[tag + "_rays"]
            print(name, tag, "gpu", c.closest_rays, c.shadow_rays, c.shadow_occluded, c.nan_samples, "ref", [int(x) for x in r])
