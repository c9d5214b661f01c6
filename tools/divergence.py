"""Where do the fp32 and the fp64 render of the SAME samples part ways? Renders the centre crop of the headline frame
(1080p room, tess 64) in both precisions with the decision log on (fcgpu_set_decision_log) and compares the discrete
decisions of every path, vertex by vertex: the first vertex where they differ is the path's divergence, classified by
what differs there. Pixels are then compared as the parity gate does. usage: divergence.py [spp] [crop_w crop_h]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
S = importlib.import_module("ray-tracing_amd.scenes")
R = importlib.import_module("ray-tracing_amd.renderer")
A = importlib.import_module("ray-tracing_amd._abi")

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
tw, th = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (240, 135)
scene = S.room(W, H, tess=64)
tile = ((W - tw) // 2, (H - th) // 2, tw, th)
out = {}
for prec in ("f64", "f32"):
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision=prec)
        ctx.upload_scene(scene)
        ctx.set_sampler(0)
        ctx.set_decision_log(True)
        ctx.render_pass(0, spp, A.INTEGRATOR_MIS, 10, True, tile)
        hits, flags = ctx.read_decision_log()
        rgb, n = ctx.read_film()
        out[prec] = (hits, flags, rgb[tile[1]:tile[1] + th, tile[0]:tile[0] + tw])
(h64, f64, img64), (h32, f32, img32) = out["f64"], out["f32"]
bounces, paths = h64.shape
OCC = 16
path_bits = ~np.uint32(OCC | 8)  # the path itself: shaded / sampled / transmitted / lobe type (not the shadow ray's fate)
hit_diff = h64 != h32
flag_diff = (f64 & path_bits) != (f32 & path_bits)
diff = hit_diff | flag_diff
first = np.where(diff.any(axis=0), diff.argmax(axis=0), -1)
diverged = first >= 0
print(f"{paths} paths ({tw}x{th} x {spp} spp), {bounces} vertices each; paths whose decisions differ between fp32 and fp64: "
      f"{diverged.sum()} = {diverged.mean():.2e}")
idx = np.flatnonzero(diverged)
fb = first[idx]
a, b = h64[fb, idx], h32[fb, idx]
fa, fbits = f64[fb, idx], f32[fb, idx]
MISS, DEAD = 0xFFFFFFFE, 0xFFFFFFFD
both_hit = (a < DEAD) & (b < DEAD)
cls = np.full(len(idx), "other", dtype=object)
hd = a != b
cls[hd & both_hit & ((a & 0x3FFFFFFF) != (b & 0x3FFFFFFF))] = "closest hit: another primitive"
cls[hd & both_hit & ((a & 0x3FFFFFFF) == (b & 0x3FFFFFFF))] = "lobe pick (same primitive)"
cls[hd & ~both_hit] = "hit on one side, miss / end on the other"
same = ~hd
cls[same & (((fa ^ fbits) >> 8) & 15 != 0)] = "lobe type (same hit)"
cls[same & (((fa ^ fbits) >> 8) & 15 == 0) & (((fa ^ fbits) & 2) != 0)] = "BSDF sample succeeds on one side only"
cls[same & (((fa ^ fbits) >> 8) & 15 == 0) & (((fa ^ fbits) & 2) == 0) & (((fa ^ fbits) & 4) != 0)] = "reflect / transmit pick (Fresnel)"
print("first divergence by cause:")
for name in sorted(set(cls), key=lambda k: -(cls == k).sum()):
    m = cls == name
    per_vertex = np.bincount(fb[m] + 1, minlength=bounces + 1)[1:]
    print(f"  {name:44s} {m.sum():7d} = {m.sum() / paths:.2e} of the paths; by vertex 1..{bounces}: {per_vertex.tolist()}")
# the primitive kinds involved in 'another primitive'
m = cls == "closest hit: another primitive"
if m.any():
    ka, kb = (a[m] >> 28) & 3, (b[m] >> 28) & 3
    names = {0: "triangle", 1: "sphere", 2: "plane"}
    pairs = {}
    for x, y in zip(ka, kb):
        pairs[(names[int(x)], names[int(y)])] = pairs.get((names[int(x)], names[int(y)]), 0) + 1
    print("  'another primitive' by kinds (fp64, fp32):", pairs)
    tri = (ka == 0) & (kb == 0)
    d = np.abs((a[m][tri] & 0x0FFFFFFF).astype(np.int64) - (b[m][tri] & 0x0FFFFFFF).astype(np.int64))
    print(f"  triangle / triangle cases whose records are neighbours in leaf order (|index difference| <= 2): {(d <= 2).mean():.2f}")
# shadow rays: cast / occluded differences on paths that did NOT diverge before that vertex
sh = ((f64 ^ f32) & (8 | OCC)) != 0
before = (first[None, :] < 0) | (np.arange(bounces)[:, None] <= first[None, :])
sh_only = sh & before & ~diff
print(f"visibility differences at vertices the two paths still share: {sh_only.sum()} = {sh_only.sum() / max(1, ((f64 & 8) != 0).sum()):.2e} of the shadow rays "
      f"(cast on one side only: {(((f64 ^ f32) & 8) != 0)[sh_only].sum()}, occluded on one side only: {((((f64 ^ f32) & OCC) != 0) & (((f64 ^ f32) & 8) == 0))[sh_only].sum()})")
# pixels
floor = 1e-3 * float(np.abs(img64).mean())
rel = np.abs(img32 - img64).max(axis=2) / np.maximum(np.abs(img64).max(axis=2), floor)
pix_of_path = np.arange(paths) % (tw * th)
has_div = np.zeros(tw * th, bool)
has_div[pix_of_path[diverged]] = True
has_sh = np.zeros(tw * th, bool)
has_sh[pix_of_path[sh_only.any(axis=0)]] = True
bad = (rel > 1e-2).ravel()
print(f"pixels beyond 1e-2 (fp32 against fp64): {bad.sum()} = {bad.mean():.4f}; of these {(bad & has_div).sum()} hold a diverged path, "
      f"{(bad & ~has_div & has_sh).sum()} only a visibility difference, {(bad & ~has_div & ~has_sh).sum()} neither")
print(f"pixels with a diverged path: {has_div.mean():.4f}; median rel err of the pixels without any difference: {np.median(rel.ravel()[~has_div & ~has_sh]):.2e}, "
      f"their maximum: {rel.ravel()[~has_div & ~has_sh].max():.2e}")
