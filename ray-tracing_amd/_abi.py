"""ctypes mirror of include/fcgpu.h (the C-ABI boundary). Plain data only; no compute here."""
import ctypes as C

c_double3 = C.c_double * 3
c_double2 = C.c_double * 2

SURFACE_SPHERE, SURFACE_PLANE, SURFACE_MESH = 0, 1, 2
MATERIAL_DIFFUSE, MATERIAL_MIRROR, MATERIAL_GLASS = 0, 1, 2
MATERIAL_PLASTIC, MATERIAL_STANDARD, MATERIAL_TRANSMISSION = 3, 4, 5
ENV_NONE, ENV_CONST, ENV_TEXTURE = 0, 1, 2
IMAGE_R8, IMAGE_RGB8, IMAGE_SRGB8, IMAGE_RGB32 = 0, 1, 2, 3
IMAGE_FORMAT_IDS = {"r8": 0, "rgb8": 1, "srgb8": 2, "rgb32": 3}
FILTER_BOX, FILTER_BILINEAR = 0, 1
TEXTURE_IMAGE, TEXTURE_CHECKER = 0, 1
PRECISION_F32, PRECISION_F64 = 0, 1
INTEGRATOR_BSDF, INTEGRATOR_MIS = 0, 1
BVH_HOST_SAH, BVH_DEVICE_LBVH, BVH_HOST_REFERENCE = 0, 1, 2
OFFSET_DEFAULT, OFFSET_NORMAL, OFFSET_ALONG_RAY = 0, 1, 2

BVH_WIDE8 = 0x100
E_INVALID, E_HIP, E_NOSCENE, E_LIMIT, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6


class Surface(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("has_normals", C.c_int32), ("has_uvs", C.c_int32), ("reserved", C.c_int32),
        ("position", c_double3), ("rotation", c_double3), ("scale", c_double3),
        ("radius", C.c_double), ("size", c_double2),
        ("vertex_offset", C.c_uint32), ("vertex_count", C.c_uint32),
        ("index_offset", C.c_uint32), ("index_count", C.c_uint32),
    ]


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("reserved", C.c_int32),
        ("color_a", c_double3), ("color_b", c_double3), ("roughness", c_double2),
        ("ior", C.c_double), ("metalness", C.c_double),
        ("tex_color_a", C.c_int32), ("tex_color_b", C.c_int32), ("tex_roughness", C.c_int32),
        ("tex_metalness", C.c_int32), ("tex_ior", C.c_int32), ("tex_normal", C.c_int32),
    ]


class Texture(C.Structure):
    _fields_ = [("image", C.c_int32), ("filter", C.c_int32), ("kind", C.c_int32), ("reserved", C.c_int32),
                ("color_a", c_double3), ("color_b", c_double3), ("scale", C.c_double)]


class Medium(C.Structure):
    _fields_ = [
        ("priority", C.c_int32), ("reserved", C.c_int32),
        ("ior", C.c_double), ("color", c_double3), ("density", C.c_double),
    ]


class Entity(C.Structure):
    _fields_ = [
        ("surface", C.c_int32), ("material", C.c_int32), ("medium", C.c_int32), ("emissive", C.c_int32),
        ("emission_color", c_double3), ("emission_strength", C.c_double),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", c_double3), ("rotation", c_double3),
        ("fov", C.c_double), ("lens_radius", C.c_double), ("focus_distance", C.c_double),
        ("width", C.c_int32), ("height", C.c_int32),
    ]


class Image(C.Structure):
    _fields_ = [("format", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32), ("pixels", C.c_void_p)]


class EnvTexture(C.Structure):
    _fields_ = [
        ("image", C.c_int32), ("filter", C.c_int32), ("integral_sample_count", C.c_int32),
        ("distribution_width", C.c_int32), ("distribution_height", C.c_int32), ("reserved", C.c_int32),
        ("strength", C.c_double), ("rotation", c_double3),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("surfaces", C.POINTER(Surface)), ("materials", C.POINTER(Material)),
        ("media", C.POINTER(Medium)), ("entities", C.POINTER(Entity)),
        ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
        ("uvs", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
        ("surface_count", C.c_uint32), ("material_count", C.c_uint32),
        ("medium_count", C.c_uint32), ("entity_count", C.c_uint32),
        ("vertex_total", C.c_uint32), ("index_total", C.c_uint32),
        ("env_kind", C.c_int32), ("reserved", C.c_int32),
        ("env_color", c_double3),
        ("camera", Camera),
        ("images", C.POINTER(Image)), ("image_count", C.c_uint32), ("reserved2", C.c_uint32),
        ("env_texture", EnvTexture),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32), ("reserved3", C.c_uint32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("precision", C.c_int32), ("sort_materials", C.c_int32), ("epsilon", C.c_double),
        ("max_paths_in_flight", C.c_uint32), ("bvh_builder", C.c_uint32),
        ("origin_offset", C.c_uint32), ("bvh_width", C.c_uint32),
    ]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class Ray(C.Structure):
    _fields_ = [("origin", c_double3), ("direction", c_double3), ("t_max", C.c_double)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_double), ("b0", C.c_double), ("b1", C.c_double),
                ("prim", C.c_uint32), ("reserved", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("shadow_occluded", C.c_uint64), ("nan_samples", C.c_uint64), ("stack_overflows", C.c_uint64),
        ("bvh_nodes", C.c_uint64), ("primitives", C.c_uint64), ("kernel_launches", C.c_uint64),
        ("reserved", C.c_uint64 * 7), ("medium_overflows", C.c_uint64), ("medium_deep_pushes", C.c_uint64), ("reduce_bytes_copied", C.c_uint64),
    ]


class BsdfQuery(C.Structure):
    _fields_ = [("wo", c_double3), ("wi", c_double3), ("u_lobe", C.c_double), ("u1", c_double2), ("u2", c_double2)]


class BsdfResult(C.Structure):
    _fields_ = [("f", c_double3), ("pdf", C.c_double), ("s_wi", c_double3), ("s_value", c_double3), ("s_pdf", C.c_double),
                ("lobe", C.c_int32), ("is_delta", C.c_int32), ("s_ok", C.c_int32), ("reserved", C.c_int32)]


class SceneStats(C.Structure):
    _fields_ = [("primitives", C.c_uint64), ("triangles", C.c_uint64), ("binary_nodes", C.c_uint64), ("wide_nodes", C.c_uint64),
                ("max_stack", C.c_uint64), ("lobe_mask", C.c_uint64), ("reserved", C.c_uint64 * 2)]


class StageTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8)]


STAGE_NAMES = ["generate", "extend", "shade", "shadow", "accumulate", "sort"]

# numpy dtypes with the same layout as Ray / Hit (for bulk ray sets)
import numpy as _np

ray_dtype = _np.dtype([("origin", "<f8", 3), ("direction", "<f8", 3), ("t_max", "<f8")])
hit_dtype = _np.dtype([("t", "<f8"), ("b0", "<f8"), ("b1", "<f8"), ("prim", "<u4"), ("reserved", "<u4")])
bsdf_query_dtype = _np.dtype([("wo", "<f8", 3), ("wi", "<f8", 3), ("u_lobe", "<f8"), ("u1", "<f8", 2), ("u2", "<f8", 2)])
bsdf_result_dtype = _np.dtype([("f", "<f8", 3), ("pdf", "<f8"), ("s_wi", "<f8", 3), ("s_value", "<f8", 3), ("s_pdf", "<f8"),
                               ("lobe", "<i4"), ("is_delta", "<i4"), ("s_ok", "<i4"), ("reserved", "<i4")])
assert ray_dtype.itemsize == C.sizeof(Ray) and hit_dtype.itemsize == C.sizeof(Hit)
assert bsdf_query_dtype.itemsize == C.sizeof(BsdfQuery) and bsdf_result_dtype.itemsize == C.sizeof(BsdfResult)

# every symbol include/fcgpu.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "fcgpu_create", "fcgpu_destroy", "fcgpu_last_error", "fcgpu_set_options", "fcgpu_upload_scene",
    "fcgpu_set_sampler", "fcgpu_set_sampler_stratified", "fcgpu_render_pass", "fcgpu_trace_closest", "fcgpu_trace_any",
    "fcgpu_trace_closest_device", "fcgpu_trace_any_device", "fcgpu_clear_film", "fcgpu_read_film",
    "fcgpu_film_device_ptr", "fcgpu_add_sample_count", "fcgpu_get_counters", "fcgpu_reset_counters",
    "fcgpu_sync", "fcgpu_set_profiling", "fcgpu_get_stage_times", "fcgpu_stream", "fcgpu_export_image", "fcgpu_version",
    "fcgpu_check_scene", "fcgpu_render_bands", "fcgpu_reduce_film", "fcgpu_bsdf_table", "fcgpu_trace_closest_noalloc", "fcgpu_set_decision_log", "fcgpu_read_decision_log", "fcgpu_device_count",
    "fcgpu_save_film", "fcgpu_restore_film",
]
