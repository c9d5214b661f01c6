"""Host-side mirror of the reference's renderer interface over the C-ABI (include/fcgpu.h).

`GpuRenderer` is shaped like `fc::renderer` (src/renderer/renderer.hpp:20-141): construct with
the scene, integrator parameters and sampler (seed + sample count), `run()`, then
`export_image(name)` / `film()`. All compute happens in libfcgpu.so (hand-written HIP for
gfx950); there is NO CPU fallback: loading fails loudly when the library is missing.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfcgpu.so")
_lib = None


class FcgpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"fcgpu error {code}: {message}")
        self.code = code


def load_library():
    """Loads libfcgpu.so once and declares every prototype of include/fcgpu.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FCGPU_LIBRARY", LIB_PATH)  # development override: A/B runs of two builds
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    P = C.POINTER
    vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    protos = {
        "fcgpu_create": (i32, [i32, P(vp)]),
        "fcgpu_destroy": (i32, [vp]),
        "fcgpu_last_error": (C.c_char_p, [vp]),
        "fcgpu_set_options": (i32, [vp, P(A.Options)]),
        "fcgpu_upload_scene": (i32, [vp, P(A.SceneDesc)]),
        "fcgpu_set_sampler": (i32, [vp, u64]),
        "fcgpu_set_sampler_stratified": (i32, [vp, u64, u32]),
        "fcgpu_render_pass": (i32, [vp, P(A.Tile), u32, u32, i32, i32, i32]),
        "fcgpu_trace_closest": (i32, [vp, u64, vp, vp]),
        "fcgpu_trace_any": (i32, [vp, u64, vp, vp]),
        "fcgpu_trace_closest_device": (i32, [vp, u64, vp, vp]),
        "fcgpu_trace_any_device": (i32, [vp, u64, vp, vp]),
        "fcgpu_trace_closest_noalloc": (i32, [vp, u64, vp, vp]),
        "fcgpu_device_count": (i32, [P(i32)]),
        "fcgpu_set_decision_log": (i32, [vp, i32]),
        "fcgpu_read_decision_log": (i32, [vp, P(u32), P(u64), vp, vp]),
        "fcgpu_clear_film": (i32, [vp]),
        "fcgpu_read_film": (i32, [vp, vp, P(u64)]),
        "fcgpu_save_film": (i32, [vp, vp, P(u64)]),
        "fcgpu_restore_film": (i32, [vp, vp, u64]),
        "fcgpu_film_device_ptr": (i32, [vp, P(vp), P(u64)]),
        "fcgpu_add_sample_count": (i32, [vp, C.c_int64]),
        "fcgpu_get_counters": (i32, [vp, P(A.Counters)]),
        "fcgpu_reset_counters": (i32, [vp]),
        "fcgpu_sync": (i32, [vp]),
        "fcgpu_set_profiling": (i32, [vp, i32]),
        "fcgpu_get_stage_times": (i32, [vp, P(A.StageTimes)]),
        "fcgpu_stream": (i32, [vp, P(vp)]),
        "fcgpu_export_image": (i32, [vp, C.c_char_p]),
        "fcgpu_version": (i32, []),
        "fcgpu_check_scene": (i32, [P(A.SceneDesc), u32, P(A.SceneStats)]),
        "fcgpu_render_bands": (i32, [vp, u32, u32, u32, u32, u32, i32, i32, i32]),
        "fcgpu_reduce_film": (i32, [P(vp), i32, i32]),
        "fcgpu_bsdf_table": (i32, [vp, P(A.Material), vp, vp, vp, vp, C.c_double, C.c_double, u64, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check_scene(scene, bvh_builder="host", bvh_width=4):
    """Validates `scene` and builds its hierarchy on the host, no device needed (fcgpu_check_scene): SceneStats."""
    lib = load_library()
    desc = scene.desc()
    st = A.SceneStats()
    builder = {"host": A.BVH_HOST_SAH, "reference": A.BVH_HOST_REFERENCE}[bvh_builder] | (A.BVH_WIDE8 if bvh_width == 8 else 0)
    rc = lib.fcgpu_check_scene(C.byref(desc), builder, C.byref(st))
    if rc != 0:
        raise FcgpuError(rc, (lib.fcgpu_last_error(None) or b"").decode())
    return st


class _DeviceArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class GpuContext:
    """One fcgpu_ctx = one GPU, one HIP stream. Single-owner."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.fcgpu_create(device, C.byref(h))
        if rc != 0:
            raise FcgpuError(rc, self.lib.fcgpu_last_error(None).decode())
        self.h = h
        self.device = device
        self.width = self.height = 0
        self._scene = None

    def _check(self, rc):
        if rc != 0:
            raise FcgpuError(rc, self.lib.fcgpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.fcgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_options(self, precision="f32", epsilon=0.0, max_paths_in_flight=0, sort_materials=True, bvh_builder="host",
                    origin_offset="default", bvh_width=None):
        """bvh_builder: "host" = SAH build on the host (three-axis split search), "device" = linear BVH built on the GPU,
        "reference" = host SAH build with the reference's own split search (its tree)."""
        o = A.Options(precision=A.PRECISION_F64 if precision == "f64" else A.PRECISION_F32,
                      sort_materials=int(sort_materials), epsilon=epsilon, max_paths_in_flight=max_paths_in_flight,
                      bvh_builder={"host": A.BVH_HOST_SAH, "device": A.BVH_DEVICE_LBVH, "reference": A.BVH_HOST_REFERENCE}[bvh_builder],
                      origin_offset={"default": A.OFFSET_DEFAULT, "normal": A.OFFSET_NORMAL, "along_ray": A.OFFSET_ALONG_RAY}[origin_offset],
                      # development knob: FCGPU_BVH_WIDTH=8 runs everything that does not say otherwise on the 8-wide hierarchy
                      bvh_width=int(os.environ.get("FCGPU_BVH_WIDTH", "0")) if bvh_width is None else bvh_width)
        self._check(self.lib.fcgpu_set_options(self.h, C.byref(o)))

    def upload_scene(self, scene):
        desc = scene.desc()
        self._check(self.lib.fcgpu_upload_scene(self.h, C.byref(desc)))
        self._scene = scene
        self.width, self.height = scene.camera.width, scene.camera.height

    def set_sampler(self, seed, stratified=None):
        """random_sampler-like stream, or with `stratified=sample_count` the stratified_sampler{sample_count, seed} analogue."""
        if stratified is None:
            self._check(self.lib.fcgpu_set_sampler(self.h, seed))
        else:
            self._check(self.lib.fcgpu_set_sampler_stratified(self.h, seed, stratified))

    def render_pass(self, sample_begin, sample_count, integrator=A.INTEGRATOR_MIS, max_path_length=10, visible_env=True,
                    tile=None):
        t = A.Tile(*tile) if tile is not None else None
        self._check(self.lib.fcgpu_render_pass(self.h, C.byref(t) if t is not None else None, sample_begin, sample_count,
                                               integrator, max_path_length, int(visible_env)))

    def render_bands(self, band_rows, band_first, band_stride, sample_begin, sample_count, integrator=A.INTEGRATOR_MIS,
                     max_path_length=10, visible_env=True):
        """Tile sharding: the bands b of `band_rows` film rows with b % band_stride == band_first, as one pass."""
        self._check(self.lib.fcgpu_render_bands(self.h, band_rows, band_first, band_stride, sample_begin, sample_count, integrator,
                                                max_path_length, int(visible_env)))

    def bsdf_table(self, material, queries, ng=(0, 1, 0), tangent=(1, 0, 0), ns=(0, 1, 0), bitangent=(0, 0, 1), eta_a=1.0, eta_b=1.0):
        """Device-side bsdf evaluate / pdf_wi / sample_wi of a constant material at a surface frame (parity hook)."""
        q = np.ascontiguousarray(queries, dtype=A.bsdf_query_dtype)
        out = np.zeros(len(q), dtype=A.bsdf_result_dtype)
        vecs = [np.ascontiguousarray(v, dtype=np.float64) for v in (ng, tangent, ns, bitangent)]
        self._check(self.lib.fcgpu_bsdf_table(self.h, C.byref(material), *[v.ctypes.data for v in vecs], eta_a, eta_b, len(q),
                                              q.ctypes.data, out.ctypes.data))
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=A.ray_dtype)
        hits = np.zeros(len(rays), dtype=A.hit_dtype)
        self._check(self.lib.fcgpu_trace_closest(self.h, len(rays), rays.ctypes.data, hits.ctypes.data))
        return hits

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=A.ray_dtype)
        out = np.zeros(len(rays), dtype=np.uint8)
        self._check(self.lib.fcgpu_trace_any(self.h, len(rays), rays.ctypes.data, out.ctypes.data))
        return out

    def trace_closest_device(self, n, rays_ptr, hits_ptr):
        self._check(self.lib.fcgpu_trace_closest_device(self.h, n, rays_ptr, hits_ptr))

    def set_decision_log(self, enabled):
        self._check(self.lib.fcgpu_set_decision_log(self.h, int(enabled)))

    def read_decision_log(self):
        """(hit_words[bounces, paths], flag_words[bounces, paths]) of the last pass rendered with the log enabled."""
        b, n = C.c_uint32(), C.c_uint64()
        self._check(self.lib.fcgpu_read_decision_log(self.h, C.byref(b), C.byref(n), None, None))
        hits = np.zeros((b.value, n.value), np.uint32)
        flags = np.zeros((b.value, n.value), np.uint32)
        self._check(self.lib.fcgpu_read_decision_log(self.h, None, None, hits.ctypes.data, flags.ctypes.data))
        return hits, flags

    def trace_closest_noalloc(self, n, rays_ptr, hits_ptr):
        """fcgpu_trace_closest_noalloc: like trace_closest_device, scratch in the library's static device array."""
        self._check(self.lib.fcgpu_trace_closest_noalloc(self.h, n, rays_ptr, hits_ptr))

    def trace_any_device(self, n, rays_ptr, out_ptr):
        self._check(self.lib.fcgpu_trace_any_device(self.h, n, rays_ptr, out_ptr))

    def clear_film(self):
        self._check(self.lib.fcgpu_clear_film(self.h))

    def read_film(self):
        """(rgb_sum[H,W,3] float64, sample_count): the operands of renderer::export_image's division."""
        rgb = np.zeros((self.height, self.width, 3), np.float64)
        n = C.c_uint64(0)
        self._check(self.lib.fcgpu_read_film(self.h, rgb.ctypes.data, C.byref(n)))
        return rgb, n.value

    def save_film(self):
        """(raw_sum[H,W,3] float64 as the device holds it, sample_count): the checkpoint of a render between passes."""
        raw = np.zeros((self.height, self.width, 3), np.float64)
        n = C.c_uint64(0)
        self._check(self.lib.fcgpu_save_film(self.h, raw.ctypes.data, C.byref(n)))
        return raw, n.value

    def restore_film(self, raw, sample_count):
        """Puts a checkpoint (save_film) back: the next render_pass continues that render."""
        raw = np.ascontiguousarray(raw, np.float64)
        if raw.shape != (self.height, self.width, 3):
            raise ValueError(f"checkpoint of shape {raw.shape} for a {self.width}x{self.height} film")
        self._check(self.lib.fcgpu_restore_film(self.h, raw.ctypes.data, int(sample_count)))

    def film_device_view(self):
        """Object exposing __cuda_array_interface__ over the device film ([H, W, 3] float64, unscaled)."""
        p, b = C.c_void_p(), C.c_uint64()
        self._check(self.lib.fcgpu_film_device_ptr(self.h, C.byref(p), C.byref(b)))
        return _DeviceArray(p.value, (self.height, self.width, 3), "<f8", self)

    def add_sample_count(self, delta):
        self._check(self.lib.fcgpu_add_sample_count(self.h, delta))

    def counters(self):
        c = A.Counters()
        self._check(self.lib.fcgpu_get_counters(self.h, C.byref(c)))
        return c

    def reset_counters(self):
        self._check(self.lib.fcgpu_reset_counters(self.h))

    def sync(self):
        self._check(self.lib.fcgpu_sync(self.h))

    def set_profiling(self, enabled):
        self._check(self.lib.fcgpu_set_profiling(self.h, int(enabled)))

    def stage_times(self):
        """{stage: (device ms, timed launch groups)} since the previous call (HIP events on the stream)."""
        t = A.StageTimes()
        self._check(self.lib.fcgpu_get_stage_times(self.h, C.byref(t)))
        return {n: (t.ms[i], t.launches[i]) for i, n in enumerate(A.STAGE_NAMES)}

    def stream(self):
        s = C.c_void_p()
        self._check(self.lib.fcgpu_stream(self.h, C.byref(s)))
        return s.value

    def export_image(self, name):
        self._check(self.lib.fcgpu_export_image(self.h, name.encode()))


class GpuRenderer:
    """Drop-in shaped like fc::renderer{resolution, camera_factory, integrator, scene, worker_count, sampler}.

    resolution and camera live in the scene description; `integrator` is "mis" or "bsdf" with
    its max_path_length / visible_infinity_area_light (forward_mis_integrator.hpp:9-11);
    the sampler is the counter-based stream (seed, sample_count): `sampler="random"` like
    random_sampler{n, seed}, `sampler="stratified"` like stratified_sampler{n, seed}."""

    def __init__(self, scene, integrator="mis", max_path_length=10, visible_infinity_area_light=True, sample_count=16,
                 seed=0, device=0, precision="f32", tiles=None, sampler="random", **options):
        self.ctx = GpuContext(device)
        self.ctx.set_options(precision=precision, **options)
        self.ctx.upload_scene(scene)
        self.ctx.set_sampler(seed, stratified=sample_count if sampler == "stratified" else None)
        self.integrator = A.INTEGRATOR_MIS if integrator == "mis" else A.INTEGRATOR_BSDF
        self.max_path_length = max_path_length
        self.visible = visible_infinity_area_light
        # renderer::run_pixel renders sampler.get_sample_count() samples (renderer.hpp:159): for stratified_sampler that is
        # floor(sqrt(sample_count))^2 (stratified_sampler.hpp:15-16, 26-29), not the count it was constructed with
        self.sample_count = math.isqrt(int(sample_count)) ** 2 if sampler == "stratified" else sample_count
        self.tiles = tiles  # None = whole image; else list of (x0, y0, w, h) owned by this GPU

    def run(self, sample_begin=0, sample_count=None):
        n = self.sample_count if sample_count is None else sample_count
        for tile in (self.tiles if self.tiles is not None else [None]):
            self.ctx.render_pass(sample_begin, n, self.integrator, self.max_path_length, self.visible, tile)
        self.ctx.sync()
        self.verify()

    def verify(self):
        """gpu_renderer::verify (csrc/fc_renderer.hpp): a dropped traversal-stack or medium-stack push means a wrong image."""
        c = self.ctx.counters()
        if c.stack_overflows != 0:
            raise FcgpuError(A.E_LIMIT, "traversal stack overflowed (stack_overflows != 0): the image is not valid")
        if c.medium_overflows != 0:
            raise FcgpuError(A.E_LIMIT, "a path nested more than 9 media (medium_overflows != 0): the image is not valid")

    def film(self):
        return self.ctx.read_film()

    def image(self):
        rgb, n = self.ctx.read_film()
        return (rgb / float(n)).astype(np.float32)

    def export_image(self, name):
        self.ctx.export_image(name)

    def close(self):
        self.ctx.close()


def reduce_film(contexts, root=0):
    """fcgpu_reduce_film: films and sample counts of all contexts summed into contexts[root] (one process, several GPUs)."""
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = contexts[0].lib.fcgpu_reduce_film(arr, len(contexts), root)
    if rc != 0:
        raise FcgpuError(rc, contexts[0].lib.fcgpu_last_error(contexts[root].h if 0 <= root < len(contexts) else None).decode())


def shard_tiles(width, height, rank, world, tile=64):
    """Interleaved tile sharding for N GPUs (SURVEY 8e mode A): tile t goes to rank t % world after a
    stride permutation that spreads neighbouring tiles over ranks. Every pixel has exactly one owner."""
    tiles = []
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    for j in range(ty):
        for i in range(tx):
            t = j * tx + i
            if (t + j) % world == rank:
                x0, y0 = i * tile, j * tile
                tiles.append((x0, y0, min(tile, width - x0), min(tile, height - y0)))
    return tiles
