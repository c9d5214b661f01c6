"""Closest / any-hit rays per second of the trace entry points on device-resident rays (development tool).
usage: python tools/trace_perf.py [N_MILLION] [tess]   (env FCGPU_BVH_WIDTH=2: binary-node experiment)
Rays: the distribution a render casts (path vertices of the room scene -> cosine-ish directions), built on the host
from hits of uniform rays, so that origins sit on surfaces like bounce rays do."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
S = importlib.import_module("ray-tracing_amd.scenes")
Rn = importlib.import_module("ray-tracing_amd.renderer")
A = importlib.import_module("ray-tracing_amd._abi")
nm = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
tess = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(nm * 1e6)
rng = np.random.default_rng(1)
sc = S.room(1920, 1080, tess=tess)
ctx = Rn.GpuContext(0)
ctx.upload_scene(sc)
# first generation: uniform rays from inside the room; second generation: from their hit points into random directions
rays = np.zeros(n, dtype=A.ray_dtype)
rays["origin"] = rng.uniform([-11, 0.5, -9], [11, 7.5, 19], (n, 3))
d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["direction"] = d
rays["t_max"] = np.inf
hits = ctx.trace_closest(rays)
ok = hits["prim"] != 0xFFFFFFFF
p = rays["origin"] + rays["direction"] * np.where(ok, hits["t"], 1.0)[:, None]
d2 = rng.standard_normal((n, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
rays2 = np.zeros(n, dtype=A.ray_dtype)
rays2["origin"] = p - rays["direction"] * 1e-3
rays2["direction"] = d2
rays2["t_max"] = np.inf
dev = torch.device("cuda", 0)
r_dev = torch.from_numpy(rays2.view(np.uint8).reshape(-1)).to(dev)
h_dev = torch.zeros(n * A.hit_dtype.itemsize, dtype=torch.uint8, device=dev)
o_dev = torch.zeros(n, dtype=torch.uint8, device=dev)
for name, fn, out in (("closest", ctx.trace_closest_device, h_dev), ("any", ctx.trace_any_device, o_dev)):
    fn(n, r_dev.data_ptr(), out.data_ptr()); ctx.sync()
    t0 = time.time()
    for _ in range(5):
        fn(n, r_dev.data_ptr(), out.data_ptr())
    ctx.sync()
    dt = (time.time() - t0) / 5
    print(f"{name}: {n / dt / 1e9:.3f} Grays/s ({dt * 1e3:.1f} ms for {n} rays)", flush=True)
hh = np.frombuffer(h_dev.cpu().numpy().tobytes(), dtype=A.hit_dtype)
c = ctx.counters()
print("hit fraction", (hh["prim"] != 0xFFFFFFFF).mean(), "checksum prim", int(hh["prim"].astype(np.uint64).sum()), "t sum %.6f" % hh["t"][hh["prim"] != 0xFFFFFFFF].sum(),
      "occluded", int(o_dev.sum().item()), "max_stack", c.reserved[1], "nodes", c.bvh_nodes, "overflow", c.stack_overflows)
