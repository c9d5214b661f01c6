"""Development tool (CPU): node steps, child-box tests and primitive tests per ray of the 4-wide and the 8-wide hierarchy on
bounce-like rays of the room scene (origins on surfaces, uniform directions), walked the way the kernels visit them
(tools/widestats/wide_stats.cpp), and the closest-hit ids of the two hierarchies compared.
usage: python tools/wide_stats.py [N_THOUSAND] [tess] [scene]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
S = importlib.import_module("ray-tracing_amd.scenes")
A = importlib.import_module("ray-tracing_amd._abi")
lib = C.CDLL(os.path.join(ROOT, "tools", "widestats", "libwidestats.so"))
lib.wide_stats.restype = C.c_int
lib.wide_stats.argtypes = [C.POINTER(A.SceneDesc), C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

n = int(float(sys.argv[1]) * 1000) if len(sys.argv) > 1 else 200000
tess = int(sys.argv[2]) if len(sys.argv) > 2 else 64
name = sys.argv[3] if len(sys.argv) > 3 else "room"
sc = S.room(1920, 1080, tess=tess) if name == "room" else S.BUILDERS[name]()
desc = sc.desc()


def walk(width, rays):
    pk = np.zeros(len(rays), np.uint32)
    t = np.zeros(len(rays), np.float64)
    st = np.zeros(8, np.uint64)
    assert lib.wide_stats(C.byref(desc), width, len(rays), rays.ctypes.data, pk.ctypes.data, t.ctypes.data, st.ctypes.data) == 0
    return pk, t, st


rng = np.random.default_rng(1)
rays = np.zeros(n, dtype=A.ray_dtype)
lo, hi = ([-11, 0.5, -9], [11, 7.5, 19]) if name == "room" else ([-3, 0.2, -3], [3, 4, 3])
rays["origin"] = rng.uniform(lo, hi, (n, 3))
d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["direction"] = d
rays["t_max"] = np.inf
pk, t, _ = walk(4, rays)
ok = pk != 0xFFFFFFFF
d2 = rng.standard_normal((n, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
bounce = np.zeros(n, dtype=A.ray_dtype)
bounce["origin"] = rays["origin"] + rays["direction"] * np.where(ok, t, 1.0)[:, None] - rays["direction"] * 1e-3
bounce["direction"] = d2
bounce["t_max"] = np.inf
out = {}
for width in (4, 8):
    pk, t, st = walk(width, bounce)
    out[width] = (pk, t)
    print(f"{width}-wide: {int(st[4])} nodes, builder's worst-case stack {int(st[5])}, deepest seen {int(st[3])}; per ray: node steps {st[0] / n:.2f}, "
          f"child-box tests {st[1] / n:.2f}, primitive tests {st[2] / n:.2f}, 16-byte node requests {st[0] * (4 if width == 4 else 5) / n:.1f}")
    if width == 4:
        print(f"        node steps on the first 85 / 341 nodes in level order (what an LDS copy of the top of the tree would serve): {st[6] / st[0]:.3f} / {st[7] / st[0]:.3f}")
same = out[4][0] == out[8][0]
near = np.abs(out[4][1] - out[8][1]) <= 1e-9 * np.maximum(1.0, out[4][1])
print(f"closest-hit ids equal on {same.mean():.6f} of the rays; of the others {int((~same & near).sum())} are ties in distance, {int((~same & ~near).sum())} are not")
