"""Development probe: which paths do the fp32 pixels of the C4 crop that differ WITHOUT any differing decision go through?
usage: c4_probe.py [min_roughness]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common as T

z, scene, tile = T.load_crop("c4_mask")
if len(sys.argv) > 1:
    import ast
    kw = ast.literal_eval(str(z["kwargs"]))
    kw["min_roughness"] = float(sys.argv[1])
    scene = T.S.mask_scene(**kw)
img64, _, _, h64, f64 = T.render_crop(scene, z, tile, "f64", decision_log=True)
img32, _, _, h32, f32 = T.render_crop(scene, z, tile, "f32", decision_log=True)
npix = tile[2] * tile[3]
d = T.divergence_table(h64, f64, img64, h32, f32, img32, npix)
floor = 1e-3 * float(np.abs(img64).mean())
rel = (np.abs(img32 - img64).max(axis=2) / np.maximum(np.abs(img64).max(axis=2), floor)).ravel()
diff = (h64 != h32) | ((f64 & ~np.uint32(24)) != (f32 & ~np.uint32(24)))
sh = ((f64 ^ f32) & 24) != 0
pix = np.arange(h64.shape[1]) % npix
dirty = np.zeros(npix, bool)
dirty[pix[diff.any(axis=0) | sh.any(axis=0)]] = True
bad = (rel > 1e-2) & ~dirty
print("bad clean pixels", bad.sum())
ys, xs = np.divmod(np.flatnonzero(bad), tile[2])
print("x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
# lobe type shaded at each vertex for the paths of bad pixels vs all
key = (f64 >> 8) & 15
shaded = (f64 & 1) != 0
names = ["lambert", "ggx_refl", "spec_refl", "ggx_glass", "spec_glass", "ggx_trans", "spec_trans", "final"]
for label, sel in (("bad clean pixels", bad[pix]), ("all pixels", np.ones_like(pix, bool))):
    print(label, "paths", sel.sum())
    for v in range(3):
        k = key[v][sel & shaded[v]]
        print("  vertex", v + 1, {names[i]: int((k == i).sum()) for i in range(8) if (k == i).any()}, "shadow cast", int(((f64[v] & 8) != 0)[sel].sum()), "unoccluded", int((((f64[v] & 8) != 0) & ((f64[v] & 16) == 0))[sel].sum()))
print("rel err quantiles of bad clean:", np.quantile(rel[bad], [0.5, 0.9, 1.0]))
# per-pixel abs diff relative to pixel value, and pixel brightness vs mean
b = np.flatnonzero(bad)[:10]
for i in b:
    print(" pixel", i % tile[2], i // tile[2], "f64", img64.reshape(-1, 3)[i], "f32", img32.reshape(-1, 3)[i])
