"""Development aid: shadow-ray count of the GPU vs the reference's any-hit count, per film fixture."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from common import A, FILM_FIXTURES, R, film_stratified, load_film
for name in FILM_FIXTURES:
    z, scene = load_film(name)
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f64")
        ctx.upload_scene(scene)
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        for tag, integ in (("mis", A.INTEGRATOR_MIS),):
            ctx.clear_film(); ctx.reset_counters()
            ctx.render_pass(0, int(z["spp"]), integ, int(z["max_path_length"]), True)
            c = ctx.counters()
            rays = z[tag + "_rays"]
            print(f"{name:24s} shadow gpu {c.shadow_rays:8d} ref {int(rays[1]):8d} diff {int(c.shadow_rays) - int(rays[1]):5d} occluded gpu {c.shadow_occluded} ref {int(rays[2])}", flush=True)
