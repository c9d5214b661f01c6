"""Development probe: per-sample radiance of single pixels of the C4 crop in fp64 and fp32, with the path's decision log."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as T
from common import A, R

z, scene, tile = T.load_crop("c4_mask")
pixels = [(94, 2), (116, 6), (109, 1), (101, 7)]
ctxs = {}
for prec in ("f64", "f32"):
    ctx = R.GpuContext(0)
    ctx.set_options(precision=prec)
    ctx.upload_scene(scene)
    ctx.set_sampler(0)
    ctx.set_decision_log(True)
    ctxs[prec] = ctx
names = ["lambert", "ggx_refl", "spec_refl", "ggx_glass", "spec_glass", "ggx_trans", "spec_trans", "final"]
for (px, py) in pixels:
    x, y = tile[0] + px, tile[1] + py
    for s in range(8):
        vals = {}
        for prec, ctx in ctxs.items():
            ctx.clear_film()
            ctx.render_pass(s, 1, A.INTEGRATOR_MIS, 10, True, (x, y, 1, 1))
            rgb, n = ctx.read_film()
            h, f = ctx.read_decision_log()
            vals[prec] = (rgb[y, x].copy(), h[:, 0].copy(), f[:, 0].copy())
        a, b = vals["f64"][0], vals["f32"][0]
        rel = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
        if rel > 1e-4:
            h, f = vals["f64"][1], vals["f64"][2]
            desc = []
            for v in range(10):
                if h[v] == 0xFFFFFFFF:
                    break
                if h[v] >= 0xFFFFFFFD:
                    desc.append("miss" if h[v] == 0xFFFFFFFE else "dead")
                else:
                    desc.append(f"{names[(f[v] >> 8) & 15]}(flags {f[v] & 255:#x}, lobe {h[v] >> 30}, tri {h[v] & 0x0FFFFFFF})")
            same = np.array_equal(vals["f64"][1], vals["f32"][1]) and np.array_equal(vals["f64"][2], vals["f32"][2])
            print(f"pixel ({px},{py}) sample {s}: f64 {a} f32 {b} rel {rel:.3g} log equal {same}: " + " -> ".join(desc))
