"""First-contact GPU diagnostics: trace parity + small render parity against the real reference
(oracle/_ref/libfcref.so). Development tool, not part of the product."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("ray-tracing_amd")
S = importlib.import_module("ray-tracing_amd.scenes")
Rn = importlib.import_module("ray-tracing_amd.renderer")
A = importlib.import_module("ray-tracing_amd._abi")
from oracle import pyref

def random_rays(n, lo, hi, seed=1):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=A.ray_dtype)
    r["origin"] = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["direction"] = d
    r["t_max"] = np.inf
    return r

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "room"
    kw = eval(sys.argv[2]) if len(sys.argv) > 2 else dict(width=160, height=90, tess=8)
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    sc = S.BUILDERS[name](**kw)
    ref = pyref.reference().scene(sc)
    rays = random_rays(200000, [-11, 0.5, -9], [11, 7.5, 19])
    h_ref = ref.trace_closest(rays)
    a_ref = ref.trace_any(rays)
    for prec in ("f64", "f32"):
        ctx = Rn.GpuContext(0)
        ctx.set_options(precision=prec)
        ctx.upload_scene(sc)
        c = ctx.counters()
        print(prec, "nodes", c.bvh_nodes, "prims", c.primitives, "max_stack", c.reserved[1], "binary nodes", c.reserved[2])
        t0 = time.time(); h = ctx.trace_closest(rays); t1 = time.time()
        a = ctx.trace_any(rays)
        mism = h["prim"] != h_ref["prim"]
        hit = (h_ref["prim"] != 0xFFFFFFFF) & ~mism
        terr = np.abs(h["t"][hit] - h_ref["t"][hit]) / np.maximum(1, h_ref["t"][hit])
        print(f"  closest: prim mismatch {mism.sum()}/{len(rays)}  max rel t err {terr.max():.3e}  any mismatch {(a != a_ref).sum()}  ({t1-t0:.3f}s)")
        print("  overflow", ctx.counters().stack_overflows)
        for integ, iname in ((A.INTEGRATOR_MIS, "mis"), (A.INTEGRATOR_BSDF, "bsdf")):
            ctx.clear_film(); ctx.reset_counters()
            t0 = time.time()
            ctx.render_pass(0, spp, integ, 10, True)
            rgb, n = ctx.read_film()
            t1 = time.time()
            rr, rn, st = ref.render(spp, integrator=integ, threads=8)
            c = ctx.counters()
            img, rimg = rgb / n, rr / rn
            rel = np.abs(img - rimg).max(axis=2) / np.maximum(rimg.max(axis=2), 1e-3)
            print(f"  {iname}: gpu {t1-t0:.3f}s n={n} ref n={rn}; mean gpu {img.mean():.6f} ref {rimg.mean():.6f}; "
                  f"rel err median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} max {rel.max():.2e}; frac>1e-3 {(rel>1e-3).mean():.4f}")
            print(f"     rays gpu closest {c.closest_rays} shadow {c.shadow_rays} occ {c.shadow_occluded} nan {c.nan_samples} | ref closest {st.closest_rays} any {st.any_rays} occ {st.any_occluded} nan {st.nan_samples}")
        ctx.close()

main()
