"""Development tool (GPU): the 8-wide hierarchy against the 4-wide one: closest / any hits on the golden ray sets, films of a few
scenes in fp64 (must be equal up to exact-tie hits) and the bench workload's stage times.
usage: python tools/wide8_check.py [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as T
from common import A, R, S

for name in T.RAY_FIXTURES:
    z, scene = T.load_rays(name)
    out = {}
    for w in (4, 8):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision="f64", bvh_width=w)
            ctx.upload_scene(scene)
            out[w] = (ctx.trace_closest(z["rays"]), ctx.trace_any(z["rays"]), ctx.counters())
    h4, a4, c4 = out[4]
    h8, a8, c8 = out[8]
    print(f"rays_{name}: closest ids equal {(h4['prim'] == h8['prim']).mean():.6f}, vs reference {(h8['prim'] == z['bvh_prim']).mean():.6f}, any equal {(a4 == a8).mean():.6f}, "
          f"t max diff {np.nanmax(np.abs(np.where(h4['prim'] == h8['prim'], h4['t'] - h8['t'], 0.0))):.2e}, overflow {c8.stack_overflows}, nodes {c4.bvh_nodes} / {c8.bvh_nodes}, stack {c4.reserved[1]} / {c8.reserved[1]}")
for name in ("room_t4", "zoo", "nested_rough", "mask", "glass"):
    z, scene = T.load_film(name)
    for prec, tol in (("f64", 1e-9), ("f32", None)):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=prec, bvh_width=8)
            ctx.upload_scene(scene)
            ctx.set_sampler(int(z["seed"]), stratified=T.film_stratified(z))
            ctx.render_pass(0, int(z["spp"]), A.INTEGRATOR_MIS, int(z["max_path_length"]), True)
            rgb, n = ctx.read_film()
            c = ctx.counters()
        rel = T.pixel_rel_err(rgb, z["mis_rgb_sum"])
        print(f"film_{name} {prec} 8-wide vs reference: max rel {rel.max():.2e} median {np.median(rel):.2e} within 1e-2 {(rel <= 1e-2).mean():.4f}; closest rays {int(c.closest_rays)} vs {int(z['mis_rays'][0])}, overflow {c.stack_overflows}")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = S.room(1920, 1080, tess=64)
for w in (4, 8, 4, 8):
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32", bvh_width=w)
        ctx.upload_scene(sc)
        ctx.render_pass(0, 2); ctx.sync()
        ctx.set_profiling(True); ctx.stage_times(); ctx.reset_counters(); ctx.sync()
        t0 = time.time(); ctx.render_pass(2, spp); ctx.sync(); dt = time.time() - t0
        st = ctx.stage_times(); c = ctx.counters()
        rays = c.closest_rays + c.shadow_rays
        rgb, n = ctx.read_film()
    print(f"width {w}: {1920*1080*spp/dt/1e6:.1f} Msamples/s | " + " ".join(f"{k} {v[0]:.1f}" for k, v in st.items()) + f" | node steps/ray {c.reserved[3]/rays:.2f} prims/ray {c.reserved[4]/rays:.2f} mean {(rgb/n).mean():.6f} overflow {c.stack_overflows}", flush=True)
