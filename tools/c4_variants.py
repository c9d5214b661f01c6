"""Development probe: per-PATH fp32 error on the C4 crop (1 spp passes, so a pixel is one path) with one feature dropped at a time."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as T
from common import A, R

z, _, tile = T.load_crop("c4_mask")
kw = ast.literal_eval(str(z["kwargs"]))
for variant in (sys.argv[1:] or ["", "nonormal", "notex", "constenv", "boxenv", "pinhole", "notex constenv", "nonormal constenv pinhole"]):
    scene = T.S.mask_scene(**kw, variant=variant)
    out = {}
    for prec in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=prec)
            ctx.upload_scene(scene)
            ctx.set_sampler(0)
            ctx.render_pass(0, 1, A.INTEGRATOR_MIS, 10, True, tile)
            rgb, n = ctx.read_film()
            x0, y0, w, h = tile
            out[prec] = rgb[y0:y0 + h, x0:x0 + w]
    a, b = out["f64"], out["f32"]
    on_mask = a.max(axis=2) > 0
    rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(a).max(axis=2), 1e-3 * np.abs(a).mean())
    hit = rel[on_mask]
    print(f"variant '{variant}': paths {hit.size}, rel err median {np.median(hit):.2e} p90 {np.quantile(hit, 0.9):.2e} p99 {np.quantile(hit, 0.99):.2e} "
          f"beyond 1e-2: {(hit > 1e-2).mean():.4f} beyond 1e-3: {(hit > 1e-3).mean():.4f}")
