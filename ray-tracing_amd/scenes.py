"""Scene descriptions (plain data) for the C-ABI of include/fcgpu.h.

The reference's room scene (src/example_scenes.hpp:187-263) loads three meshes (`walls`,
`lights_1`, `lights_2`) that are NOT in the reference repository (assets/ is git-ignored), so
`room()` authors a documented procedural stand-in for them; spheres, materials, the medium, the
emitter strength and the camera are the reference's values verbatim.
"""
import ctypes as C
import math

import numpy as np

from . import _abi as A


class Scene:
    """Mutable builder; `desc()` freezes it into a ctypes fcgpu_scene_desc (arrays kept alive)."""

    def __init__(self):
        self.surfaces, self.materials, self.media, self.entities = [], [], [], []
        self._pos, self._nrm, self._uv, self._idx = [], [], [], []
        self._nvert = 0
        self._nidx = 0
        self.camera = A.Camera()
        self.env_kind = A.ENV_NONE
        self.env_color = (0.0, 0.0, 0.0)
        self.images = []          # (format id, contiguous pixel array of shape (height, width, channels))
        self.textures = []        # A.Texture: image textures usable as material parameters
        self.env_texture = A.EnvTexture()
        self._keep = None

    # ---- surfaces
    def sphere(self, position, radius, rotation=(0, 0, 0)):
        s = A.Surface(kind=A.SURFACE_SPHERE, radius=radius)
        s.position[:] = position
        s.rotation[:] = rotation
        s.scale[:] = (1, 1, 1)
        self.surfaces.append(s)
        return len(self.surfaces) - 1

    def plane(self, position, size, rotation=(0, 0, 0)):
        s = A.Surface(kind=A.SURFACE_PLANE)
        s.position[:] = position
        s.rotation[:] = rotation
        s.scale[:] = (1, 1, 1)
        s.size[:] = size
        self.surfaces.append(s)
        return len(self.surfaces) - 1

    def mesh(self, positions, indices, normals=None, uvs=None, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        assert indices.size % 3 == 0 and (indices.size == 0 or indices.max() < len(positions))
        s = A.Surface(kind=A.SURFACE_MESH, has_normals=int(normals is not None), has_uvs=int(uvs is not None))
        s.position[:] = position
        s.rotation[:] = rotation
        s.scale[:] = scale
        s.vertex_offset, s.vertex_count = self._nvert, len(positions)
        s.index_offset, s.index_count = self._nidx, indices.size
        self._pos.append(positions)
        self._nrm.append(np.zeros_like(positions) if normals is None else
                         np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3))
        self._uv.append(np.zeros((len(positions), 2), np.float32) if uvs is None else
                        np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 2))
        self._idx.append(indices)
        self._nvert += len(positions)
        self._nidx += indices.size
        self.surfaces.append(s)
        return len(self.surfaces) - 1

    # ---- materials / media
    def material(self, kind, color_a=(0, 0, 0), color_b=(0, 0, 0), roughness=(0, 0), ior=1.0, metalness=0.0, **tex):
        """`tex`: tex_color_a / tex_color_b / tex_roughness / tex_metalness / tex_ior = handles from texture()
        that replace the constant of the same name (see fcgpu_material)."""
        m = A.Material(kind=kind, ior=ior, metalness=metalness, **tex)
        m.color_a[:] = color_a
        m.color_b[:] = color_b
        m.roughness[:] = roughness
        self.materials.append(m)
        return len(self.materials) - 1

    def diffuse(self, reflectance):
        return self.material(A.MATERIAL_DIFFUSE, color_a=reflectance)

    def mirror(self, reflectance, roughness):
        return self.material(A.MATERIAL_MIRROR, color_a=reflectance, roughness=roughness)

    def glass(self, reflectance, transmittance, roughness):
        return self.material(A.MATERIAL_GLASS, color_a=reflectance, color_b=transmittance, roughness=roughness)

    def plastic(self, diffuse, specular, roughness, ior):
        return self.material(A.MATERIAL_PLASTIC, color_a=diffuse, color_b=specular, roughness=roughness, ior=ior)

    def standard(self, base_color, metalness, roughness, ior):
        return self.material(A.MATERIAL_STANDARD, color_a=base_color, roughness=(roughness, roughness), ior=ior,
                             metalness=metalness)

    def transmission(self, transmittance, roughness):
        return self.material(A.MATERIAL_TRANSMISSION, color_b=transmittance, roughness=roughness)

    def medium(self, priority, ior, color, density):
        m = A.Medium(priority=priority, ior=ior, density=density)
        m.color[:] = color
        self.media.append(m)
        return len(self.media) - 1

    def entity(self, surface, material, medium=-1, emission=None, strength=0.0):
        e = A.Entity(surface=surface, material=material, medium=medium, emissive=int(emission is not None),
                     emission_strength=strength)
        e.emission_color[:] = emission if emission is not None else (0, 0, 0)
        self.entities.append(e)
        return len(self.entities) - 1

    def set_camera(self, position, rotation, fov, width, height, lens_radius=0.0, focus_distance=0.0):
        c = self.camera
        c.position[:] = position
        c.rotation[:] = rotation
        c.fov, c.lens_radius, c.focus_distance = fov, lens_radius, focus_distance
        c.width, c.height = width, height

    def set_env(self, color):
        self.env_kind = A.ENV_CONST
        self.env_color = tuple(color)

    def image(self, pixels, fmt):
        """An image as assets::get_image returns it: raw pixels (height, width[, channels]) in format
        'r8' | 'rgb8' | 'srgb8' | 'rgb32' (see assets.py). Returns its index."""
        from . import assets as AS
        dtype, ch = AS.IMAGE_FORMATS[fmt]
        px = np.ascontiguousarray(pixels, dtype=dtype)
        if px.ndim == 2:
            px = px[:, :, None]
        assert px.ndim == 3 and px.shape[2] == ch
        self.images.append((A.IMAGE_FORMAT_IDS[fmt], np.ascontiguousarray(px)))
        return len(self.images) - 1

    def texture(self, image, filter="bilinear"):
        """image_texture_2d_rgb / image_texture_2d_r over `image`; returns the handle to pass as a material's
        tex_* argument (1-based: 0 means "use the constant")."""
        self.textures.append(A.Texture(image=image, filter=A.FILTER_BILINEAR if filter == "bilinear" else A.FILTER_BOX))
        return len(self.textures)

    def checker(self, color_a, color_b, scale):
        """checker_texture_2d_rgb{color_a, color_b, scale} (rgb parameters only); returns a handle like texture()."""
        t = A.Texture(kind=A.TEXTURE_CHECKER, scale=scale)
        t.color_a[:] = color_a
        t.color_b[:] = color_b
        self.textures.append(t)
        return len(self.textures)

    def set_env_texture(self, image, strength=1.0, rotation=(0, 0, 0), filter="bilinear", integral_sample_count=4,
                        distribution_resolution=None):
        """texture_infinity_area_light{{{}, rotation}, image_texture_2d_rgb{image, filter, integral_sample_count},
        strength, distribution_resolution}; the resolution defaults to the image's (example_scenes.hpp:339-341)."""
        h, w = self.images[image][1].shape[:2]
        dw, dh = distribution_resolution if distribution_resolution is not None else (w, h)
        self.env_kind = A.ENV_TEXTURE
        t = A.EnvTexture(image=image, filter=A.FILTER_BILINEAR if filter == "bilinear" else A.FILTER_BOX,
                         integral_sample_count=integral_sample_count, distribution_width=dw, distribution_height=dh,
                         strength=strength)
        t.rotation[:] = rotation
        self.env_texture = t

    # ---- queries
    def primitive_counts(self):
        out = []
        for e in self.entities:
            s = self.surfaces[e.surface]
            out.append(s.index_count // 3 if s.kind == A.SURFACE_MESH else 1)
        return out

    def primitive_total(self):
        return sum(self.primitive_counts())

    def desc(self):
        d = A.SceneDesc()
        surf = (A.Surface * max(1, len(self.surfaces)))(*self.surfaces)
        mats = (A.Material * max(1, len(self.materials)))(*self.materials)
        meds = (A.Medium * max(1, len(self.media)))(*self.media)
        ents = (A.Entity * max(1, len(self.entities)))(*self.entities)
        pos = np.concatenate(self._pos) if self._pos else np.zeros((1, 3), np.float32)
        nrm = np.concatenate(self._nrm) if self._nrm else np.zeros((1, 3), np.float32)
        uv = np.concatenate(self._uv) if self._uv else np.zeros((1, 2), np.float32)
        idx = np.concatenate(self._idx) if self._idx else np.zeros((3,), np.uint32)
        pos, nrm, uv, idx = (np.ascontiguousarray(a) for a in (pos, nrm, uv, idx))
        d.surfaces, d.materials, d.media, d.entities = surf, mats, meds, ents
        d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
        d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
        d.uvs = uv.ctypes.data_as(C.POINTER(C.c_float))
        d.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
        d.surface_count, d.material_count = len(self.surfaces), len(self.materials)
        d.medium_count, d.entity_count = len(self.media), len(self.entities)
        d.vertex_total, d.index_total = self._nvert, self._nidx
        d.env_kind = self.env_kind
        d.env_color[:] = self.env_color
        d.camera = self.camera
        imgs = (A.Image * max(1, len(self.images)))()
        for k, (fmt, px) in enumerate(self.images):
            imgs[k] = A.Image(format=fmt, width=px.shape[1], height=px.shape[0], pixels=px.ctypes.data)
        d.images = imgs
        d.image_count = len(self.images)
        d.env_texture = self.env_texture
        texs = (A.Texture * max(1, len(self.textures)))(*self.textures)
        d.textures = texs
        d.texture_count = len(self.textures)
        self._keep = (surf, mats, meds, ents, pos, nrm, uv, idx, imgs, texs)
        return d


# --------------------------------------------------------------------------- mesh helpers
def grid_quad(origin, du, dv, nu, nv):
    """(nu x nv) quads spanning origin + s*du + t*dv, s,t in [0,1]; normal = cross(du, dv)."""
    origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    p = origin + s[..., None] * du + t[..., None] * dv
    p = p.reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).reshape(-1)
    b = a + (nv + 1)
    c = b + 1
    d = a + 1
    tris = np.stack([a, b, c, a, c, d], axis=1).reshape(-1)
    return p.astype(np.float32), tris.astype(np.uint32)


def merge(parts):
    pos, idx, off = [], [], 0
    for p, i in parts:
        pos.append(p)
        idx.append(i + off)
        off += len(p)
    return np.concatenate(pos), np.concatenate(idx).astype(np.uint32)


def box_inward(lo, hi, n, faces="xXyYzZ"):
    """Axis-aligned box whose faces (subset `faces`) face the INSIDE; each face n x n quads."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ex, ey, ez = (hi - lo) * np.eye(3)
    parts = []
    if "y" in faces:  # floor, normal +y = cross(dz, dx)
        parts.append(grid_quad(lo, ez, ex, n, n))
    if "Y" in faces:  # ceiling, normal -y = cross(dx, dz)
        parts.append(grid_quad(lo + ey, ex, ez, n, n))
    if "x" in faces:  # x = lo wall, normal +x = cross(dy, dz)
        parts.append(grid_quad(lo, ey, ez, n, n))
    if "X" in faces:  # x = hi wall, normal -x = cross(dz, dy)
        parts.append(grid_quad(lo + ex, ez, ey, n, n))
    if "z" in faces:  # z = lo wall, normal +z = cross(dx, dy)
        parts.append(grid_quad(lo, ex, ey, n, n))
    if "Z" in faces:  # z = hi wall, normal -z = cross(dy, dx)
        parts.append(grid_quad(lo + ez, ey, ex, n, n))
    return merge(parts)


def box_outward(lo, hi, faces="xXyYzZ"):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ex, ey, ez = (hi - lo) * np.eye(3)
    parts = []
    if "y" in faces:
        parts.append(grid_quad(lo, ex, ez, 1, 1))
    if "Y" in faces:
        parts.append(grid_quad(lo + ey, ez, ex, 1, 1))
    if "x" in faces:
        parts.append(grid_quad(lo, ez, ey, 1, 1))
    if "X" in faces:
        parts.append(grid_quad(lo + ex, ey, ez, 1, 1))
    if "z" in faces:
        parts.append(grid_quad(lo, ey, ex, 1, 1))
    if "Z" in faces:
        parts.append(grid_quad(lo + ez, ex, ey, 1, 1))
    return merge(parts)


# --------------------------------------------------------------------------- the room
ROOM_LO, ROOM_HI = (-12.0, 0.0, -10.0), (12.0, 8.0, 20.0)
ROOM_LIGHTS = [(x, z) for z in (-2.0, 10.0) for x in (-7.5, -2.5, 2.5, 7.5)]

CENTER_MATERIALS = ("rough_glass", "smooth_glass", "diffuse", "plastic", "rough_mirror", "smooth_mirror",
                    "smooth_plastic", "standard", "transmission")


def room_meshes(tess=64):
    """The three stand-in meshes of the room by their reference asset names -> (positions, indices)."""
    wp, wi = box_inward(ROOM_LO, ROOM_HI, tess)
    fixtures = merge([box_outward((x - 0.4, 7.9, z - 0.4), (x + 0.4, 8.0, z + 0.4), faces="xXyzZ")
                      for x, z in ROOM_LIGHTS])
    quads = merge([grid_quad((x - 0.25, 7.88, z - 0.25), (0.5, 0, 0), (0, 0, 0.5), 1, 1) for x, z in ROOM_LIGHTS])
    return {"walls": (wp, wi), "lights_1": fixtures, "lights_2": quads}


def room(width=1920, height=1080, tess=64, center="rough_glass", env=None, assets_root=None):
    """Stand-in for example_scenes.hpp:187-263.

    Verbatim from the reference: the three spheres (:218-242), their materials, the medium
    (priority 1, ior 1.45, colour (0.6,0.2,0.8), density 0.5), emitter (1,1,1) x 250 (:215),
    wall albedo 0.8 (:195), fixture albedo 0.1 (:203,:212), camera (:250).
    Invented (the meshes are absent): `walls` = closed box ROOM_LO..ROOM_HI, each face
    tess x tess quads (12*tess^2 triangles); `lights_2` = eight 0.5 x 0.5 downward quads at
    y = 7.88; `lights_1` = open fixture boxes (no top face) around them, y in [7.9, 8].
    `center` swaps the middle sphere's material for the material sweep (BASELINE config 3).
    `assets_root`: load `walls`, `lights_1`, `lights_2` from <assets_root>/assets/*.mesh (the
    reference's asset format, see assets.py) instead of generating them.
    """
    sc = Scene()
    if assets_root is None:
        meshes = room_meshes(tess)
    else:
        from . import assets as AS
        meshes = {}
        for name in ("walls", "lights_1", "lights_2"):
            m = AS.read_mesh(AS.mesh_path(assets_root, name))
            meshes[name] = (m["positions"], m["indices"])
    walls = sc.mesh(*meshes["walls"])
    lights_1 = sc.mesh(*meshes["lights_1"])
    lights_2 = sc.mesh(*meshes["lights_2"])

    sc.entity(walls, sc.diffuse((0.8, 0.8, 0.8)))
    sc.entity(lights_1, sc.diffuse((0.1, 0.1, 0.1)))
    sc.entity(lights_2, sc.diffuse((0.1, 0.1, 0.1)), emission=(1.0, 1.0, 1.0), strength=250.0)

    sc.entity(sc.sphere((4.2, 2.01, 0.0), 2.0), sc.diffuse((0.2, 0.4, 0.8)))

    mid = sc.sphere((0.0, 2.01, 0.0), 2.0)
    med = -1
    if center == "rough_glass":
        mat = sc.glass((1, 1, 1), (1, 1, 1), (0.3, 0.3))
        med = sc.medium(1, 1.45, (0.6, 0.2, 0.8), 0.5)
    elif center == "smooth_glass":
        mat = sc.glass((1, 1, 1), (1, 1, 1), (0.0, 0.0))
        med = sc.medium(1, 1.45, (0.6, 0.2, 0.8), 0.5)
    elif center == "diffuse":
        mat = sc.diffuse((0.8, 0.6, 0.2))
    elif center == "plastic":
        mat = sc.plastic((0.8, 0.6, 0.2), (1, 1, 1), (0.15, 0.15), 1.45)
    elif center == "smooth_plastic":
        mat = sc.plastic((0.8, 0.6, 0.2), (1, 1, 1), (0.0, 0.0), 1.45)
    elif center == "rough_mirror":
        mat = sc.mirror((0.8, 0.8, 0.8), (0.2, 0.2))
    elif center == "smooth_mirror":
        mat = sc.mirror((0.8, 0.8, 0.8), (0.0, 0.0))
    elif center == "standard":
        mat = sc.standard((0.8, 0.5, 0.3), 0.6, 0.35, 1.45)
    elif center == "transmission":
        mat = sc.transmission((0.9, 0.9, 0.9), (0.25, 0.25))
        med = sc.medium(1, 1.3, (0.2, 0.6, 0.8), 0.3)
    else:
        raise ValueError(center)
    sc.entity(mid, mat, medium=med)

    sc.entity(sc.sphere((-4.2, 2.01, 0.0), 2.0), sc.mirror((0.8, 0.4, 0.2), (0.2, 0.2)))

    sc.set_camera((-7.23, 2.79, 16.37), (math.radians(-4.641), math.radians(157.552), 0.0), math.radians(35.0),
                  width, height)
    if env is not None:
        sc.set_env(env)
    return sc


def room_planes(width=512, height=288):
    """The 10-primitive stand-in SURVEY.md section 6 timed: 6 wall planes + 1 plane emitter + 3 spheres."""
    sc = Scene()
    white = sc.diffuse((0.8, 0.8, 0.8))
    lo, hi = np.array(ROOM_LO), np.array(ROOM_HI)
    cx, cy, cz = (lo + hi) / 2
    sx, sy, sz = hi - lo
    h = math.pi / 2
    sc.entity(sc.plane((cx, lo[1], cz), (sx, sz)), white)                                  # floor
    sc.entity(sc.plane((cx, hi[1], cz), (sx, sz), (math.pi, 0, 0)), white)                 # ceiling
    sc.entity(sc.plane((lo[0], cy, cz), (sy, sz), (0, 0, -h)), white)                      # x = lo
    sc.entity(sc.plane((hi[0], cy, cz), (sy, sz), (0, 0, h)), white)                       # x = hi
    sc.entity(sc.plane((cx, cy, lo[2]), (sx, sy), (h, 0, 0)), white)                       # z = lo
    sc.entity(sc.plane((cx, cy, hi[2]), (sx, sy), (-h, 0, 0)), white)                      # z = hi
    sc.entity(sc.plane((0.0, 7.9, 4.0), (3.0, 3.0), (math.pi, 0, 0)), sc.diffuse((0.1, 0.1, 0.1)),
              emission=(1, 1, 1), strength=250.0)
    sc.entity(sc.sphere((4.2, 2.01, 0.0), 2.0), sc.diffuse((0.2, 0.4, 0.8)))
    sc.entity(sc.sphere((0.0, 2.01, 0.0), 2.0), sc.glass((1, 1, 1), (1, 1, 1), (0.3, 0.3)),
              medium=sc.medium(1, 1.45, (0.6, 0.2, 0.8), 0.5))
    sc.entity(sc.sphere((-4.2, 2.01, 0.0), 2.0), sc.mirror((0.8, 0.4, 0.2), (0.2, 0.2)))
    sc.set_camera((-7.23, 2.79, 16.37), (math.radians(-4.641), math.radians(157.552), 0.0), math.radians(35.0),
                  width, height)
    return sc


def nested_media(width=256, height=256, smooth=True, env=(1.0, 1.0, 1.0)):
    """Priority-media exercise in the spirit of scene_glass (example_scenes.hpp:101-185): a glass
    shell (priority 2) holding water (priority 1) with an ice ball (priority 2) overlapping the
    water surface, over a diffuse floor, lit by a constant environment and a sphere emitter."""
    sc = Scene()
    r = (0.0, 0.0) if smooth else (0.2, 0.2)
    white = (1, 1, 1)
    sc.entity(sc.plane((0, 0, 0), (30, 30)), sc.diffuse((0.5, 0.5, 0.5)))
    sc.entity(sc.sphere((0, 2.0, 0), 2.0), sc.glass(white, white, r), medium=sc.medium(2, 1.52, (1, 1, 1), 1.0))
    sc.entity(sc.sphere((0, 1.8, 0), 1.7), sc.glass(white, white, r), medium=sc.medium(1, 1.3614, (0.2, 0.6, 0.8), 1.0))
    sc.entity(sc.sphere((0.4, 3.2, 0.2), 0.5), sc.glass(white, white, r), medium=sc.medium(2, 1.35, (0.2, 0.6, 0.8), 0.0))
    sc.entity(sc.sphere((-3.5, 1.0, 1.0), 1.0), sc.plastic((0.8, 0.1, 0.05), white, (0.2, 0.2), 1.46))
    sc.entity(sc.sphere((3.0, 5.0, -2.0), 0.7), sc.diffuse((0.1, 0.1, 0.1)), emission=(1, 0.9, 0.8), strength=30.0)
    sc.set_camera((0.5, 3.5, -9.0), (math.radians(10.0), 0.0, 0.0), math.radians(40.0), width, height)
    if env is not None:
        sc.set_env(env)
    return sc


def nested_deep(width=64, height=64, depth=9, smooth=True):
    """`depth` concentric glass spheres, each bounding its own medium, priorities alternating so that every second
    boundary is hidden by the medium around it (integrator.hpp:66-71: pushed by the re-cast) and the others refract
    (pushed when the path leaves the boundary inwards, :36-44): a path that reaches the core holds `depth` media on the
    helper's stack, the reference's capacity at depth 9 (integrator.hpp:24: vacuum + 9). depth 10 is beyond what the
    reference can hold (it writes past its array): only the device's refusal is tested there."""
    sc = Scene()
    r = (0.0, 0.0) if smooth else (0.15, 0.15)
    white = (1, 1, 1)
    sc.entity(sc.plane((0, 0, 0), (30, 30)), sc.diffuse((0.5, 0.5, 0.5)))
    prio = [3, 1, 4, 2, 5, 3, 6, 4, 7, 5, 8, 6]
    for k in range(depth):
        tint = (0.9 - 0.05 * (k % 3), 0.95 - 0.04 * ((k + 1) % 3), 0.92 - 0.05 * ((k + 2) % 3))
        sc.entity(sc.sphere((0.02 * k, 2.0, 0.01 * k), 2.0 - 0.18 * k), sc.glass(white, white, r),
                  medium=sc.medium(prio[k], 1.05 + 0.04 * k, tint, 0.3))
    sc.entity(sc.sphere((3.0, 5.0, -2.0), 0.7), sc.diffuse((0.1, 0.1, 0.1)), emission=(1, 0.9, 0.8), strength=30.0)
    sc.set_camera((0.5, 3.5, -9.0), (math.radians(10.0), 0.0, 0.0), math.radians(40.0), width, height)
    sc.set_env((0.8, 0.9, 1.0))
    return sc


def primitives_zoo(width=128, height=128):
    """All three surface kinds with rotations, a mesh with normals + uvs, every material class,
    a mesh emitter and a plane emitter -- small, for exhaustive parity tests."""
    sc = Scene()
    sc.entity(sc.plane((0, 0, 0), (16, 16)), sc.diffuse((0.7, 0.7, 0.7)))
    sc.entity(sc.plane((0, 4, -6), (16, 8), (math.radians(90), 0, 0)), sc.plastic((0.2, 0.6, 0.3), (1, 1, 1), (0.3, 0.1), 1.5))
    # an icosphere-ish bumpy mesh with vertex normals and uvs
    nu, nv = 12, 8
    th, ph = np.meshgrid(np.linspace(0, math.pi, nv + 1), np.linspace(0, 2 * math.pi, nu + 1), indexing="ij")
    rad = 1.0 + 0.08 * np.sin(3 * ph) * np.sin(2 * th)
    p = np.stack([rad * np.sin(th) * np.cos(ph), rad * np.cos(th), rad * np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    nrm = p / np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-9)
    uv = np.stack([ph / (2 * math.pi), th / math.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1)
    b, c, d = a + (nu + 1), a + (nu + 2), a + 1
    tris = np.stack([a, b, c, a, c, d], 1).reshape(-1)
    sc.entity(sc.mesh(p, tris, normals=nrm, uvs=uv, position=(-2.5, 1.3, 0.5), rotation=(0.3, 0.5, 0.1), scale=(1.2, 1.0, 0.9)),
              sc.standard((0.9, 0.6, 0.2), 0.7, 0.3, 1.45))
    sc.entity(sc.sphere((0.5, 1.0, 1.0), 1.0, (0.2, 0.7, 0.4)), sc.glass((1, 1, 1), (1, 1, 1), (0.0, 0.0)),
              medium=sc.medium(1, 1.5, (0.9, 0.4, 0.3), 0.4))
    sc.entity(sc.sphere((3.0, 1.2, -0.5), 1.2), sc.mirror((0.9, 0.9, 0.9), (0.0, 0.0)))
    sc.entity(sc.sphere((2.2, 0.6, 2.4), 0.6), sc.transmission((0.9, 0.9, 1.0), (0.2, 0.2)),
              medium=sc.medium(1, 1.33, (0.3, 0.5, 0.9), 0.2))
    bp, bi = box_outward((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5))
    sc.entity(sc.mesh(bp, bi, position=(-0.5, 0.0, 3.0), rotation=(0, 0.6, 0)), sc.mirror((0.7, 0.8, 0.9), (0.25, 0.1)))
    qp, qi = grid_quad((-1.0, 5.0, -1.0), (2.0, 0, 0), (0, 0, 2.0), 2, 2)   # faces down
    sc.entity(sc.mesh(qp, qi), sc.diffuse((0.1, 0.1, 0.1)), emission=(1.0, 0.95, 0.9), strength=20.0)
    sc.entity(sc.plane((-4.0, 3.0, 2.0), (1.5, 1.5), (0, 0, math.radians(-110))), sc.diffuse((0.1, 0.1, 0.1)),
              emission=(0.4, 0.6, 1.0), strength=15.0)
    sc.set_camera((0.0, 3.2, 9.5), (math.radians(12.0), math.radians(180.0), 0.0), math.radians(42.0), width, height)
    return sc


def sky_image(width=32, height=16, fmt="rgb32"):
    """A small procedural latitude-longitude environment: horizon-bright sky, dark ground, a sun blob
    (values up to ~60 in rgb32; tone-mapped into bytes for the 8-bit formats)."""
    v, u = np.meshgrid((np.arange(height) + 0.5) / height, (np.arange(width) + 0.5) / width, indexing="ij")
    up = np.cos(v * math.pi)
    sky = np.where(up > 0, 0.35 + 0.9 * (1 - up) ** 3, 0.06)[:, :, None] * np.array([0.55, 0.75, 1.0])
    sun = 60.0 * np.exp(-(((u - 0.3) * 2) ** 2 + (v - 0.28) ** 2) / 0.004)[:, :, None] * np.array([1.0, 0.85, 0.6])
    img = sky + sun
    if fmt == "rgb32":
        return img.astype(np.float32)
    byte = np.clip(255.0 * (img / (1.0 + img)) ** (1 / 2.2), 0, 255).astype(np.uint8)
    return byte[:, :, 0] if fmt == "r8" else byte


def env_texture_scene(width=96, height=64, fmt="rgb32", filter="bilinear", lens=0.05, emitter=False):
    """Open scene under a texture_infinity_area_light (the lighting of scene_mask,
    example_scenes.hpp:339-341, incl. its thin-lens camera :350): diffuse floor, mirror and glass
    spheres, a mesh box with a rough standard material; optionally a small emitter so that the
    light pick has two candidates."""
    sc = Scene()
    sc.entity(sc.plane((0, 0, 0), (30, 30)), sc.diffuse((0.6, 0.6, 0.6)))
    sc.entity(sc.sphere((-1.6, 1.0, 0.0), 1.0), sc.mirror((0.9, 0.9, 0.9), (0.1, 0.1)))
    sc.entity(sc.sphere((1.2, 0.8, 0.8), 0.8), sc.glass((1, 1, 1), (1, 1, 1), (0.0, 0.0)), medium=sc.medium(1, 1.5, (0.8, 0.9, 0.5), 0.3))
    bp, bi = box_outward((-0.6, 0.0, -0.6), (0.6, 1.2, 0.6))
    sc.entity(sc.mesh(bp, bi, position=(0.2, 0.0, -2.2), rotation=(0, 0.5, 0)), sc.standard((0.8, 0.3, 0.2), 0.3, 0.4, 1.45))
    if emitter:
        sc.entity(sc.sphere((0.0, 3.5, 1.0), 0.3), sc.diffuse((0.1, 0.1, 0.1)), emission=(1.0, 0.8, 0.6), strength=8.0)
    img = sc.image(sky_image(32, 16, fmt), fmt)
    sc.set_env_texture(img, strength=1.5, rotation=(0.2, 1.1, 0.0), filter=filter, integral_sample_count=4 if filter == "bilinear" else 2)
    sc.set_camera((0.5, 2.2, 7.0), (math.radians(10.0), math.radians(183.0), 0.0), math.radians(40.0), width, height,
                  lens_radius=lens, focus_distance=7.0)
    return sc


def pattern_image(width, height, fmt, seed=0, lo=0.0, hi=1.0):
    """A deterministic blocky test pattern (checker + per-cell noise) in any image format; rgb32 values lie
    in [lo, hi], byte formats use the full 0..255 range (incl. exact 0 and 255)."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    checker = ((x // 2 + y // 2) % 2).astype(np.float64)
    ch = 1 if fmt == "r8" else 3
    val = np.clip(0.15 + 0.7 * checker[:, :, None] + rng.uniform(-0.15, 0.15, (height, width, ch)), 0.0, 1.0)
    if fmt == "rgb32":
        return (lo + (hi - lo) * val).astype(np.float32)
    byte = np.round(255.0 * val).astype(np.uint8)
    byte[0::3, 0::4] = 0      # exact zeros / ones: a roughness texel of 0 switches to the specular lobe,
    byte[1::3, 2::4] = 255    # a metalness texel of 1 drops the dielectric lobes (standard_material.hpp:49,89)
    return byte[:, :, 0] if fmt == "r8" else byte


def textured_scene(width=96, height=64):
    """Image textures on material parameters (image_texture_2d_rgb / image_texture_2d_r): every material class
    with at least one textured parameter, on a plane (plane uvs), a mesh with uvs, a mesh without (default uvs)
    and spheres (uv = (0, 0)); box and bilinear filters; all four image formats."""
    sc = Scene()
    checker = sc.texture(sc.image(pattern_image(16, 16, "rgb8", 1), "rgb8"), "box")
    soft = sc.texture(sc.image(pattern_image(8, 8, "srgb8", 2), "srgb8"), "bilinear")
    hdr = sc.texture(sc.image(pattern_image(8, 4, "rgb32", 3, 0.2, 0.95), "rgb32"), "bilinear")
    ior_map = sc.texture(sc.image(pattern_image(4, 4, "rgb32", 4, 1.3, 1.8), "rgb32"), "box")
    rough_map = sc.texture(sc.image(pattern_image(12, 6, "r8", 5), "r8"), "box")
    metal_map = sc.texture(sc.image(pattern_image(6, 6, "r8", 6), "r8"), "bilinear")
    metal_box = sc.texture(sc.image(pattern_image(5, 7, "r8", 7), "r8"), "box")

    sc.entity(sc.plane((0, 0, 0), (16, 16)), sc.material(A.MATERIAL_DIFFUSE, tex_color_a=checker))
    sc.entity(sc.plane((0, 4, -6), (16, 8), (math.radians(90), 0, 0)),
              sc.material(A.MATERIAL_PLASTIC, color_b=(1, 1, 1), roughness=(0.3, 0.1), tex_color_a=soft, tex_ior=ior_map))
    nu, nv = 12, 8
    th, ph = np.meshgrid(np.linspace(0, math.pi, nv + 1), np.linspace(0, 2 * math.pi, nu + 1), indexing="ij")
    p = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    uv = np.stack([ph / (2 * math.pi), th / math.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1)
    b, c, d = a + (nu + 1), a + (nu + 2), a + 1
    tris = np.stack([a, c, b, a, d, c], 1).reshape(-1)  # wound so that geometric normals point outwards like the vertex normals
    sc.entity(sc.mesh(p, tris, normals=p.copy(), uvs=uv, position=(-2.5, 1.3, 0.5), rotation=(0.3, 0.5, 0.1), scale=(1.3, 1.3, 1.3)),
              sc.material(A.MATERIAL_STANDARD, ior=1.45, tex_color_a=hdr, tex_metalness=metal_map, tex_roughness=rough_map))
    sc.entity(sc.mesh(p, tris, uvs=uv, position=(2.6, 1.1, 1.6), scale=(1.1, 1.1, 1.1)),
              sc.material(A.MATERIAL_STANDARD, color_a=(0.8, 0.7, 0.3), roughness=(0.25, 0.25), tex_metalness=metal_box, tex_ior=ior_map))
    bp, bi = box_outward((-0.7, 0.0, -0.7), (0.7, 1.4, 0.7))
    sc.entity(sc.mesh(bp, bi, position=(0.3, 0.02, 2.6), rotation=(0, 0.6, 0)),  # off the floor: no coplanar faces (exact t ties)
              sc.material(A.MATERIAL_GLASS, roughness=(0.0, 0.0), tex_color_a=soft, tex_color_b=hdr), medium=sc.medium(1, 1.5, (0.9, 0.6, 0.4), 0.3))
    sc.entity(sc.sphere((0.4, 1.0, -1.5), 1.0, (0.2, 0.7, 0.4)), sc.material(A.MATERIAL_MIRROR, roughness=(0.15, 0.05), tex_color_a=hdr))
    sc.entity(sc.mesh(bp, bi, position=(4.6, 0.03, -1.5), rotation=(0, -0.4, 0), scale=(1.2, 1.5, 1.2)),
              sc.material(A.MATERIAL_TRANSMISSION, roughness=(0.2, 0.2), tex_color_b=checker), medium=sc.medium(1, 1.33, (0.3, 0.5, 0.9), 0.2))
    qp, qi = grid_quad((-1.0, 5.0, -1.0), (2.0, 0, 0), (0, 0, 2.0), 2, 2)
    sc.entity(sc.mesh(qp, qi), sc.diffuse((0.1, 0.1, 0.1)), emission=(1.0, 0.95, 0.9), strength=20.0)
    sc.set_env((0.3, 0.35, 0.45))
    sc.set_camera((0.0, 3.2, 9.5), (math.radians(12.0), math.radians(180.0), 0.0), math.radians(42.0), width, height)
    return sc


def normal_map_image(width, height, fmt="rgb8", strength=1.0, seed=0):
    """Tangent-space normal map of a bumpy height field, encoded as the reference's materials decode it
    (rgb * 2 - 1, then blue becomes the shading-frame up axis: diffuse_material.hpp:23-26). A quarter of the
    rgb32 texels are exactly flat (0.5, 0.5, 1), which takes normal_mapping's skip path."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    h = np.sin(2 * math.pi * 2 * x / width) * np.cos(2 * math.pi * 1.5 * y / height) + 0.35 * rng.standard_normal((height, width))
    gx = strength * (np.roll(h, -1, 1) - np.roll(h, 1, 1))
    gy = strength * (np.roll(h, -1, 0) - np.roll(h, 1, 0))
    n = np.stack([-gx, -gy, np.ones_like(h)], -1)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    enc = n * 0.5 + 0.5
    if fmt == "rgb32":
        enc[0::2, 0::2] = (0.5, 0.5, 1.0)
        return enc.astype(np.float32)
    return np.round(255.0 * enc).astype(np.uint8)


def normal_mapped_scene(width=96, height=64):
    """normal_mapping<T> (bsdfs/normal_mapping.hpp) over every lobe class the reference wraps: lambertian
    (diffuse_material), specular and microfacet reflection (mirror_material), and standard_material's three
    lobes, with strong enough tilts that all 3 x 3 facet cases of evaluate / pdf / sample occur."""
    sc = Scene()
    bumps8 = sc.texture(sc.image(normal_map_image(16, 16, "rgb8", 0.9, 1), "rgb8"), "bilinear")
    bumps32 = sc.texture(sc.image(normal_map_image(12, 8, "rgb32", 1.4, 2), "rgb32"), "box")
    waves = sc.texture(sc.image(normal_map_image(24, 12, "rgb8", 0.5, 3), "rgb8"), "bilinear")
    paint = sc.texture(sc.image(pattern_image(8, 8, "srgb8", 2), "srgb8"), "bilinear")
    metal_map = sc.texture(sc.image(pattern_image(6, 6, "r8", 6), "r8"), "bilinear")

    sc.entity(sc.plane((0, 0, 0), (16, 16)), sc.material(A.MATERIAL_DIFFUSE, color_a=(0.7, 0.7, 0.65), tex_normal=bumps8))
    sc.entity(sc.plane((0, 4, -6), (16, 8), (math.radians(90), 0, 0)),
              sc.material(A.MATERIAL_MIRROR, color_a=(0.9, 0.85, 0.8), roughness=(0.2, 0.1), tex_normal=bumps32))
    sc.entity(sc.plane((-5.5, 2.5, 0.0), (5, 8), (math.radians(90), math.radians(70), 0)),
              sc.material(A.MATERIAL_MIRROR, color_a=(0.9, 0.9, 0.9), roughness=(0.0, 0.0), tex_normal=waves))
    nu, nv = 12, 8
    th, ph = np.meshgrid(np.linspace(0, math.pi, nv + 1), np.linspace(0, 2 * math.pi, nu + 1), indexing="ij")
    p = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    uv = np.stack([ph / (2 * math.pi), th / math.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1)
    b, c, d = a + (nu + 1), a + (nu + 2), a + 1
    tris = np.stack([a, c, b, a, d, c], 1).reshape(-1)
    sc.entity(sc.mesh(p, tris, normals=p.copy(), uvs=uv, position=(-2.2, 1.3, 0.8), rotation=(0.3, 0.5, 0.1), scale=(1.3, 1.3, 1.3)),
              sc.material(A.MATERIAL_STANDARD, roughness=(0.3, 0.3), ior=1.45, metalness=0.5, tex_color_a=paint, tex_normal=bumps8))
    sc.entity(sc.mesh(p, tris, uvs=uv, position=(2.4, 1.15, 1.4), scale=(1.15, 1.15, 1.15)),
              sc.material(A.MATERIAL_STANDARD, color_a=(0.85, 0.7, 0.35), roughness=(0.0, 0.0), ior=1.5, tex_metalness=metal_map, tex_normal=waves))
    sc.entity(sc.sphere((0.2, 0.8, 2.8), 0.8), sc.material(A.MATERIAL_DIFFUSE, tex_color_a=paint, tex_normal=bumps32))
    qp, qi = grid_quad((-1.0, 5.0, -1.0), (2.0, 0, 0), (0, 0, 2.0), 2, 2)
    sc.entity(sc.mesh(qp, qi), sc.diffuse((0.1, 0.1, 0.1)), emission=(1.0, 0.95, 0.9), strength=20.0)
    sc.set_env((0.3, 0.35, 0.45))
    sc.set_camera((0.0, 3.2, 9.5), (math.radians(12.0), math.radians(180.0), 0.0), math.radians(42.0), width, height)
    return sc


def mask_scene(width=600, height=900, detail=48, texture_size=64, min_roughness=0.02, variant=""):
    """Stand-in for scene_mask (example_scenes.hpp:316-361, BASELINE config 4): one mesh with normals and uvs under
    standard_material{basecolor srgb8, metalness r8, roughness r8, ior 1.45, normal rgb8} (all bilinear), lit only by
    a texture_infinity_area_light (rgb32, bilinear, 4x4 integral samples, distribution = image resolution), thin-lens
    camera {2.367, 3.216, 6.485}, yaw 196.42 deg, fov 45 deg, lens 0.05, focus 6, 600x900 -- the reference's values.
    The mask mesh and its maps are not in the reference repository: the mesh here is a bumpy half-ellipsoid shell with
    eye openings, the maps are procedural."""
    sc = Scene()
    nu, nv = 2 * detail, detail
    th, ph = np.meshgrid(np.linspace(0.08 * math.pi, 0.92 * math.pi, nv + 1), np.linspace(0.0, math.pi, nu + 1), indexing="ij")
    bump = 1.0 + 0.06 * np.sin(5 * ph) * np.sin(4 * th) + 0.1 * np.exp(-((ph - math.pi / 2) ** 2 + (th - 1.75) ** 2) / 0.03)  # ridges + a nose
    x, y, z = 1.5 * bump * np.sin(th) * np.cos(ph), 2.0 * bump * np.cos(th), 1.0 * bump * np.sin(th) * np.sin(ph)
    p = np.stack([x, y, z], -1).reshape(-1, 3)
    uv = np.stack([ph / math.pi, (th - 0.08 * math.pi) / (0.84 * math.pi)], -1).reshape(-1, 2)
    # vertex normals from the parametrisation
    P = p.reshape(nv + 1, nu + 1, 3)
    du = np.gradient(P, axis=1)
    dv = np.gradient(P, axis=0)
    nrm = np.cross(du, dv).reshape(-1, 3)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    if np.sum(nrm * p) < 0:
        nrm = -nrm
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    u_c, v_c = (j + 0.5) / nu, (i + 0.5) / nv
    eye = ((np.abs(u_c - 0.33) < 0.07) | (np.abs(u_c - 0.67) < 0.07)) & (np.abs(v_c - 0.36) < 0.045)
    a = (i * (nu + 1) + j)[~eye].reshape(-1)
    b, c, d = a + (nu + 1), a + (nu + 2), a + 1
    t0 = np.stack([a, b, c, a, c, d], 1).reshape(-1, 3)
    # wind every triangle so that its geometric normal agrees with the vertex normals
    g = np.cross(p[t0[:, 1]] - p[t0[:, 0]], p[t0[:, 2]] - p[t0[:, 0]])
    flip = np.sum(g * nrm[t0[:, 0]], axis=1) < 0
    t0[flip] = t0[flip][:, [0, 2, 1]]
    mesh = sc.mesh(p, t0.reshape(-1), normals=nrm, uvs=uv, position=(0.67, 3.2, 0.75), rotation=(0.0, math.radians(16.42), 0.0))

    n = texture_size
    yy, xx = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
    rng = np.random.default_rng(11)
    stripes = 0.5 + 0.5 * np.sin(14 * math.pi * xx + 3 * np.sin(4 * math.pi * yy))
    base = np.stack([0.55 + 0.4 * stripes, 0.25 + 0.5 * (1 - stripes) * yy, 0.15 + 0.25 * xx], -1)
    gold = (np.abs(yy - 0.5) < 0.08) | (np.hypot(xx - 0.5, yy - 0.8) < 0.12)  # gilded band and chin
    base[gold] = (0.95, 0.75, 0.35)
    filt = "box" if "boxtex" in variant else "bilinear"
    basecolor = sc.texture(sc.image(np.round(255 * np.clip(base, 0, 1)).astype(np.uint8), "srgb8"), filt)
    metalness = sc.texture(sc.image(np.where(gold, 255, 0).astype(np.uint8), "r8"), filt)
    rough = np.clip(0.25 + 0.35 * stripes + 0.1 * rng.standard_normal((n, n)), min_roughness, 1.0)
    rough[gold] = max(0.12, min_roughness)
    roughness = sc.texture(sc.image(np.round(255 * rough).astype(np.uint8), "r8"), filt)
    normal = sc.texture(sc.image(normal_map_image(n, n, "rgb8", 0.45, 5), "rgb8"), filt)
    # `variant` (development probes only: tools/c4_variants.py): drops one feature at a time
    tex = dict(tex_color_a=basecolor, tex_metalness=metalness, tex_roughness=roughness, tex_normal=normal)
    if "nonormal" in variant:
        del tex["tex_normal"]
    if "notex" in variant:
        tex = {}
    for k in ("color_a", "metalness", "roughness"):
        if "only_" + k in variant:
            tex = {"tex_" + k: tex["tex_" + k]}
    sc.entity(mesh, sc.material(A.MATERIAL_STANDARD, color_a=(0.6, 0.4, 0.3), roughness=(0.3, 0.3), metalness=0.3, ior=1.45, **tex))

    env = sc.image(sky_image(64, 32, "rgb32"), "rgb32")
    if "constenv" in variant:
        sc.set_env((1.0, 0.9, 0.8))
    else:
        sc.set_env_texture(env, strength=1.0, rotation=(0.0, 0.0, 0.0), filter="box" if "boxenv" in variant else "bilinear", integral_sample_count=4)
    sc.set_camera((2.367, 3.216, 6.485), (0.0, math.radians(196.42), 0.0), math.radians(45.0), width, height,
                  lens_radius=0.0 if "pinhole" in variant else 0.05, focus_distance=6.0)
    return sc


def uv_sphere(nu=16, nv=10):
    """Unit sphere mesh with uvs and outward winding -> (positions, indices, uvs)."""
    th, ph = np.meshgrid(np.linspace(0, math.pi, nv + 1), np.linspace(0, 2 * math.pi, nu + 1), indexing="ij")
    p = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    uv = np.stack([ph / (2 * math.pi), th / math.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1)
    tris = np.stack([a, a + (nu + 2), a + (nu + 1), a, a + 1, a + (nu + 2)], 1).reshape(-1)
    return p.astype(np.float32), tris.astype(np.uint32), uv.astype(np.float32)


def cylinder(radius, y0, y1, n=24, top=True, bottom=True):
    """Closed (or open-ended) cylinder around the y axis with outward winding -> (positions, indices)."""
    ang = np.linspace(0, 2 * math.pi, n, endpoint=False)
    ring = np.stack([radius * np.cos(ang), np.zeros(n), radius * np.sin(ang)], -1)
    lo, hi = ring + (0, y0, 0), ring + (0, y1, 0)
    pos = [lo, hi, np.array([[0, y0, 0]]), np.array([[0, y1, 0]])]
    k = np.arange(n)
    kn = (k + 1) % n
    side = np.stack([k, n + k, n + kn, k, n + kn, kn], 1).reshape(-1)
    idx = [side]
    if bottom:
        idx.append(np.stack([np.full(n, 2 * n), k, kn], 1).reshape(-1))
    if top:
        idx.append(np.stack([np.full(n, 2 * n + 1), n + kn, n + k], 1).reshape(-1))
    return np.concatenate(pos).astype(np.float32), np.concatenate(idx).astype(np.uint32)


def material_ball_scene(width=512, height=512, ball="diffuse", detail=24):
    """Stand-in for scene_material_ball (example_scenes.hpp:28-99): a ball on a dark stand over a checker_texture_2d_rgb
    floor {0.8, 0.6, scale 10} under the texture environment; camera, materials and light are the reference's values.
    `ball` selects among the materials the reference keeps in comments: diffuse (active), glass (+ medium), mirror,
    plastic. The ball_1 / ball_2 meshes are not in the reference repository: a uv sphere and a cylinder here."""
    sc = Scene()
    cp, ci = cylinder(0.9, 0.0, 0.35, detail)
    sc.entity(sc.mesh(cp, ci), sc.diffuse((0.05, 0.05, 0.05)))
    bp, bi, buv = uv_sphere(detail, max(6, detail // 2))
    med = -1
    if ball == "glass":
        mat = sc.glass((1, 1, 1), (1, 1, 1), (0.35, 0.35))
        med = sc.medium(1, 1.45, (0.2, 0.4, 0.8), 4.0)
    elif ball == "mirror":
        mat = sc.mirror((0.8, 0.8, 0.8), (0.2, 0.2))
    elif ball == "plastic":
        mat = sc.plastic((0.8, 0.6, 0.2), (1, 1, 1), (0.15, 0.15), 1.45)
    else:
        mat = sc.diffuse((0.8, 0.6, 0.2))
    sc.entity(sc.mesh(bp, bi, normals=bp.copy(), uvs=buv, position=(0, 1.36, 0)), mat, medium=med)
    sc.entity(sc.plane((0, 0, 0), (20, 20)), sc.material(A.MATERIAL_DIFFUSE, tex_color_a=sc.checker((0.8, 0.8, 0.8), (0.6, 0.6, 0.6), 10.0)))
    sc.set_env_texture(sc.image(sky_image(64, 32, "rgb32"), "rgb32"), strength=1.0, filter="bilinear", integral_sample_count=4)
    sc.set_camera((-2.15, 5.5, -3.6), (math.radians(45), math.radians(30), 0.0), math.radians(30.0), width, height)
    return sc


def glass_scene(width=512, height=512, detail=24):
    """Stand-in for scene_glass (example_scenes.hpp:101-185): a glass (medium priority 2, ior 1.52) holding water
    (priority 1, ior 1.3614, absorbing) with an ice cube (priority 2, ior 1.35) and a plastic straw, two grey planes,
    the texture environment rotated by -15 degrees; materials, media, planes and camera are the reference's values,
    max path length 20 as in its forward-MIS line. The meshes (glass, water, ice, tube) are stand-ins: the water
    overlaps the glass wall on purpose, which the medium priorities resolve as the reference's nested meshes do."""
    sc = Scene()
    clear = lambda: sc.glass((1, 1, 1), (1, 1, 1), (0.0, 0.0))
    gp, gi = cylinder(0.55, 0.02, 1.5, detail)
    sc.entity(sc.mesh(gp, gi, position=(0.0, 0.0, 0.6)), clear(), medium=sc.medium(2, 1.52, (1.0, 1.0, 1.0), 1.0))
    wp, wi = cylinder(0.5, 0.1, 1.1, detail)
    sc.entity(sc.mesh(wp, wi, position=(0.0, 0.0, 0.6)), clear(), medium=sc.medium(1, 1.3614, (0.2, 0.6, 0.8), 4.0))
    ip, ii = box_outward((-0.15, 0.0, -0.15), (0.15, 0.3, 0.15))
    sc.entity(sc.mesh(ip, ii, position=(0.1, 0.7, 0.55), rotation=(0.3, 0.4, 0.2)), clear(), medium=sc.medium(2, 1.35, (0.2, 0.6, 0.8), 0.0))
    tp, ti = cylinder(0.04, 0.0, 1.9, max(8, detail // 2))
    sc.entity(sc.mesh(tp, ti, position=(-0.2, 0.15, 0.6), rotation=(0.0, 0.0, math.radians(-12.0))), sc.plastic((0.8, 0.1, 0.05), (1, 1, 1), (0.2, 0.2), 1.46))
    sc.entity(sc.plane((-0.05, 0.0, 0.63), (6.0, 5.0)), sc.diffuse((0.5, 0.5, 0.5)))
    sc.entity(sc.plane((0.25, 0.95, 2.89), (6.0, 5.0), (math.radians(-90.0), 0.0, 0.0)), sc.diffuse((0.5, 0.5, 0.5)))
    sc.set_env_texture(sc.image(sky_image(64, 32, "rgb32"), "rgb32"), strength=1.0, rotation=(0.0, math.radians(-15.0), 0.0), filter="bilinear",
                       integral_sample_count=4)
    sc.set_camera((-2.79, 1.846, -5.127), (math.radians(6.119), math.radians(19.319), 0.0), math.radians(30.0), width, height)
    return sc


def normals_scene(width=512, height=512, material="mirror", map_size=64):
    """scene_normals (example_scenes.hpp:267-316): a 50 x 50 plane with a normal-mapped smooth mirror (or, as in the
    reference's commented variant, a normal-mapped diffuse material) under a 10 x 1 plane emitter of strength 5, no
    environment; all numbers are the reference's. Its `wallpaper-normal` image is not in the repository: a procedural
    normal map stands in."""
    sc = Scene()
    nm = sc.texture(sc.image(normal_map_image(map_size, map_size, "rgb8", 0.6, 9), "rgb8"), "bilinear")
    if material == "diffuse":
        mat = sc.material(A.MATERIAL_DIFFUSE, color_a=(0.8, 0.8, 0.8), tex_normal=nm)
    else:
        mat = sc.material(A.MATERIAL_MIRROR, color_a=(0.8, 0.8, 0.8), roughness=(0.0, 0.0), tex_normal=nm)
    sc.entity(sc.plane((0.0, 0.0, 0.0), (50.0, 50.0)), mat)
    sc.entity(sc.plane((0.0, 1.0, 5.0), (10.0, 1.0), (math.radians(-90.0), 0.0, 0.0)), sc.diffuse((0.8, 0.8, 0.8)), emission=(1.0, 1.0, 1.0), strength=5.0)
    sc.set_camera((0.0, 18.5, -11.0), (math.radians(60.0), 0.0, 0.0), math.radians(30.0), width, height)
    return sc


def random_scene(seed=0, width=32, height=24):
    """A seeded random assembly of everything the hot path supports (differential testing): objects of all three
    surface kinds on a jittered grid (no overlaps, no coplanar faces), every material class with random constant or
    image-textured parameters and normal maps, media inside the refractive objects, an emitter, a random environment
    (none / constant / texture) and a pinhole or thin-lens camera."""
    rng = np.random.default_rng(1000 + seed)
    sc = Scene()
    col = lambda lo=0.1, hi=0.95: tuple(float(x) for x in rng.uniform(lo, hi, 3))
    fmt_rgb = lambda: ["rgb8", "srgb8", "rgb32"][int(rng.integers(3))]
    flt = lambda: ["box", "bilinear"][int(rng.integers(2))]

    def rgb_tex():
        if rng.random() < 0.25:
            return sc.checker(col(), col(), float(rng.uniform(1.5, 12.0)))
        f = fmt_rgb()
        return sc.texture(sc.image(pattern_image(int(rng.integers(3, 12)), int(rng.integers(3, 12)), f, int(rng.integers(1 << 30)), 0.05, 0.95), f), flt())

    def scalar_tex(lo, hi):
        if rng.random() < 0.5:
            return sc.texture(sc.image(pattern_image(int(rng.integers(3, 10)), int(rng.integers(3, 10)), "r8", int(rng.integers(1 << 30))), "r8"), flt())
        return sc.texture(sc.image(pattern_image(int(rng.integers(3, 10)), int(rng.integers(3, 10)), "rgb32", int(rng.integers(1 << 30)), lo, hi), "rgb32"), flt())

    def normal_tex():
        f = ["rgb8", "rgb32"][int(rng.integers(2))]
        return sc.texture(sc.image(normal_map_image(int(rng.integers(4, 14)), int(rng.integers(4, 14)), f, float(rng.uniform(0.2, 1.5)), int(rng.integers(1 << 30))), f), flt())

    def material(kinds):
        kind = kinds[int(rng.integers(len(kinds)))]
        tex = {}
        rough = (0.0, 0.0) if rng.random() < 0.35 else (float(rng.uniform(0.05, 0.6)), float(rng.uniform(0.05, 0.6)))
        if kind != A.MATERIAL_TRANSMISSION and rng.random() < 0.5:
            tex["tex_color_a"] = rgb_tex()
        if kind in (A.MATERIAL_GLASS, A.MATERIAL_PLASTIC, A.MATERIAL_TRANSMISSION) and rng.random() < 0.5:
            tex["tex_color_b"] = rgb_tex()
        if kind in (A.MATERIAL_PLASTIC, A.MATERIAL_STANDARD) and rng.random() < 0.4:
            tex["tex_ior"] = sc.texture(sc.image(pattern_image(4, 4, "rgb32", int(rng.integers(1 << 30)), 1.2, 1.9), "rgb32"), flt())
        if kind == A.MATERIAL_STANDARD:
            if rng.random() < 0.5:
                tex["tex_metalness"] = scalar_tex(0.0, 1.0)
            if rng.random() < 0.5:
                tex["tex_roughness"] = scalar_tex(0.0, 0.7)
        if kind in (A.MATERIAL_DIFFUSE, A.MATERIAL_MIRROR, A.MATERIAL_STANDARD) and rng.random() < 0.5:
            tex["tex_normal"] = normal_tex()
        metal = float(rng.choice([0.0, 1.0, rng.uniform(0.1, 0.9)]))
        return kind, sc.material(kind, color_a=col(), color_b=col(0.5, 1.0), roughness=rough if kind != A.MATERIAL_STANDARD else (rough[0], rough[0]),
                                 ior=float(rng.uniform(1.2, 1.8)), metalness=metal, **tex)

    opaque = [A.MATERIAL_DIFFUSE, A.MATERIAL_MIRROR, A.MATERIAL_PLASTIC, A.MATERIAL_STANDARD]
    everything = opaque + [A.MATERIAL_GLASS, A.MATERIAL_TRANSMISSION]
    sc.entity(sc.plane((0, 0, 0), (18, 18)), material(opaque)[1])
    if rng.random() < 0.7:
        sc.entity(sc.plane((0, 4, -7), (18, 8), (math.radians(90), 0, 0)), material(opaque)[1])

    nu, nv = 10, 6
    th, ph = np.meshgrid(np.linspace(0, math.pi, nv + 1), np.linspace(0, 2 * math.pi, nu + 1), indexing="ij")
    sp = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    suv = np.stack([ph / (2 * math.pi), th / math.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1)
    stris = np.stack([a, a + (nu + 2), a + (nu + 1), a, a + 1, a + (nu + 2)], 1).reshape(-1)
    cells = [(x, z) for x in (-4.5, -1.5, 1.5, 4.5) for z in (-3.0, 0.0, 3.0)]
    picks = rng.permutation(len(cells))[: int(rng.integers(4, 8))]
    priority = 1
    for c in picks:
        cx, cz = cells[c][0] + float(rng.uniform(-0.3, 0.3)), cells[c][1] + float(rng.uniform(-0.3, 0.3))
        size = float(rng.uniform(0.5, 1.1))
        lift = float(rng.uniform(0.02, 0.4))
        kind, mat = material(everything)
        shape = int(rng.integers(3))
        rot = tuple(float(x) for x in rng.uniform(-0.6, 0.6, 3))
        if shape == 0:
            surf = sc.sphere((cx, size + lift, cz), size, rot)
        elif shape == 1:
            bp, bi = box_outward((-size, 0.0, -size), (size, 2 * size, size))
            surf = sc.mesh(bp, bi, position=(cx, lift, cz), rotation=(0.0, rot[1], 0.0))
        else:
            surf = sc.mesh(sp, stris, normals=sp.copy() if rng.random() < 0.5 else None, uvs=suv, position=(cx, size + lift, cz), rotation=rot,
                           scale=(size, size * float(rng.uniform(0.7, 1.0)), size))
        med = -1
        if kind in (A.MATERIAL_GLASS, A.MATERIAL_TRANSMISSION):
            med = sc.medium(priority, float(rng.uniform(1.2, 1.7)), col(0.2, 0.9), float(rng.uniform(0.1, 0.8)))
            priority += 1
        sc.entity(surf, mat, medium=med)
    if seed % 2 == 1:
        # nested and overlapping media (priority stack, integrator.hpp:22-119): a refractive sphere with a smaller one
        # inside and a third one cutting through its boundary, priorities in random order
        c = (float(rng.uniform(-1.0, 1.0)), 1.6, float(rng.uniform(4.0, 5.0)))
        prios = rng.permutation(3) + priority
        outer = sc.medium(int(prios[0]), float(rng.uniform(1.3, 1.6)), col(0.3, 0.9), float(rng.uniform(0.1, 0.5)))
        inner = sc.medium(int(prios[1]), float(rng.uniform(1.1, 1.8)), col(0.3, 0.9), float(rng.uniform(0.1, 0.9)))
        cut = sc.medium(int(prios[2]), float(rng.uniform(1.2, 1.7)), col(0.3, 0.9), float(rng.uniform(0.1, 0.9)))
        smooth = (0.0, 0.0)
        sc.entity(sc.sphere(c, 1.5), sc.glass((1, 1, 1), col(0.7, 1.0), smooth if rng.random() < 0.6 else (0.2, 0.2)), medium=outer)
        sc.entity(sc.sphere((c[0] + 0.2, c[1] - 0.1, c[2]), 0.6), sc.glass((1, 1, 1), col(0.7, 1.0), smooth), medium=inner)
        sc.entity(sc.sphere((c[0] - 1.3, c[1] + 0.3, c[2] + 0.4), 0.8), sc.transmission(col(0.7, 1.0), smooth if rng.random() < 0.5 else (0.15, 0.15)), medium=cut)
    if rng.random() < 0.5:
        sc.entity(sc.sphere((float(rng.uniform(-2, 2)), 5.5, float(rng.uniform(-1, 2))), 0.5), sc.diffuse((0.1, 0.1, 0.1)), emission=col(0.5, 1.0), strength=float(rng.uniform(8, 25)))
    else:
        qp, qi = grid_quad((-1.0, 5.0, -1.0), (2.0, 0, 0), (0, 0, 2.0), 2, 2)
        sc.entity(sc.mesh(qp, qi), sc.diffuse((0.1, 0.1, 0.1)), emission=col(0.5, 1.0), strength=float(rng.uniform(8, 25)))
    env = int(rng.integers(3))
    if env == 1:
        sc.set_env(col(0.1, 0.6))
    elif env == 2:
        f = fmt_rgb()
        sc.set_env_texture(sc.image(sky_image(16, 8, f), f), strength=float(rng.uniform(0.5, 2.0)), rotation=tuple(float(x) for x in rng.uniform(-1, 1, 3)),
                           filter=flt(), integral_sample_count=int(rng.integers(1, 4)))
    lens = float(rng.uniform(0.02, 0.1)) if rng.random() < 0.4 else 0.0
    sc.set_camera((float(rng.uniform(-1, 1)), float(rng.uniform(2.5, 4.0)), 10.0), (math.radians(float(rng.uniform(8, 16))), math.radians(180.0), 0.0),
                  math.radians(float(rng.uniform(38, 50))), width, height, lens_radius=lens, focus_distance=9.0)
    return sc


BUILDERS = {"room": room, "room_planes": room_planes, "nested_media": nested_media, "nested_deep": nested_deep, "primitives_zoo": primitives_zoo,
            "env_texture_scene": env_texture_scene, "textured_scene": textured_scene, "normal_mapped_scene": normal_mapped_scene, "mask_scene": mask_scene, "random_scene": random_scene,
            "material_ball_scene": material_ball_scene, "glass_scene": glass_scene, "normals_scene": normals_scene}
