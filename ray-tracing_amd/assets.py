"""Asset files in the reference's formats (src/core/assets.cpp:91-234), host side.

Meshes, `<dir>/assets/<name>.mesh` (assets.cpp:91-160), little endian:
    header  u32 flags (bit0 normals, bit1 uvs), u32 vertex_count, u32 index_count
    then    positions f32[vertex_count][3], [normals f32[n][3]], [uvs f32[n][2]], indices u32[index_count]
Images, `<dir>/assets/images/<name>.metadata` (JSON: type "image", width, height, format r8 | rgb8 |
srgb8 | rgb32) + `<name>.asset` (raw pixels, row-major, top row first; assets.cpp:162-234).
`decode_image` applies the reference's texel decode (images/*.hpp, core/color.hpp:13-26).

The reference repository does not ship its assets (assets/ is git-ignored); `write_room_assets`
writes the stand-in meshes of scenes.room in this format so that the unmodified reference loader
(`fc::assets`) can read them -- tests/test_assets.py checks exactly that against the compiled reference.
"""
import json
import os

import numpy as np

MESH_HAS_NORMALS = 1
MESH_HAS_UVS = 2
IMAGE_FORMATS = {"r8": (np.uint8, 1), "rgb8": (np.uint8, 3), "srgb8": (np.uint8, 3), "rgb32": (np.float32, 3)}


def mesh_path(root, name):
    return os.path.join(root, "assets", name + ".mesh")


def write_mesh(path, positions, indices, normals=None, uvs=None):
    positions = np.ascontiguousarray(positions, dtype="<f4").reshape(-1, 3)
    indices = np.ascontiguousarray(indices, dtype="<u4").reshape(-1)
    if indices.size % 3 or (indices.size and int(indices.max()) >= len(positions)):
        raise ValueError("indices must be triangles over the given vertices")
    flags = 0
    blocks = [positions]
    if normals is not None:
        normals = np.ascontiguousarray(normals, dtype="<f4").reshape(-1, 3)
        if len(normals) != len(positions):
            raise ValueError("one normal per vertex")
        flags |= MESH_HAS_NORMALS
        blocks.append(normals)
    if uvs is not None:
        uvs = np.ascontiguousarray(uvs, dtype="<f4").reshape(-1, 2)
        if len(uvs) != len(positions):
            raise ValueError("one uv per vertex")
        flags |= MESH_HAS_UVS
        blocks.append(uvs)
    blocks.append(indices)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(np.array([flags, len(positions), indices.size], dtype="<u4").tobytes())
        for b in blocks:
            f.write(b.tobytes())


def read_mesh(path):
    """-> dict(positions, indices, normals | None, uvs | None); raises ValueError on a size mismatch
    (the reference terminates there, assets.cpp:127-128)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(12), dtype="<u4")
        if head.size != 3:
            raise ValueError(f"{path}: truncated header")
        flags, nv, ni = (int(x) for x in head)
        expected = 12 + 12 * nv + 4 * ni + (12 * nv if flags & MESH_HAS_NORMALS else 0) + (8 * nv if flags & MESH_HAS_UVS else 0)
        if size != expected:
            raise ValueError(f"{path}: file size {size} != {expected} expected from its header")
        pos = np.frombuffer(f.read(12 * nv), dtype="<f4").reshape(nv, 3)
        nrm = np.frombuffer(f.read(12 * nv), dtype="<f4").reshape(nv, 3) if flags & MESH_HAS_NORMALS else None
        uv = np.frombuffer(f.read(8 * nv), dtype="<f4").reshape(nv, 2) if flags & MESH_HAS_UVS else None
        idx = np.frombuffer(f.read(4 * ni), dtype="<u4")
    return {"positions": pos, "indices": idx, "normals": nrm, "uvs": uv}


def write_image(root, name, pixels, fmt):
    dtype, ch = IMAGE_FORMATS[fmt]
    pixels = np.ascontiguousarray(pixels, dtype=dtype)
    if pixels.ndim == 2:
        pixels = pixels[:, :, None]
    if pixels.ndim != 3 or pixels.shape[2] != ch:
        raise ValueError(f"{fmt} needs an array of shape (height, width, {ch})")
    d = os.path.join(root, "assets", "images")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name + ".metadata"), "w") as f:
        json.dump({"type": "image", "width": int(pixels.shape[1]), "height": int(pixels.shape[0]), "format": fmt}, f)
    with open(os.path.join(d, name + ".asset"), "wb") as f:
        f.write(pixels.astype(np.dtype(dtype).newbyteorder("<")).tobytes())


def read_image(root, name):
    """-> (raw pixel array (height, width, channels), format string)."""
    d = os.path.join(root, "assets", "images")
    with open(os.path.join(d, name + ".metadata")) as f:
        meta = json.load(f)
    if meta.get("type") != "image" or meta.get("format") not in IMAGE_FORMATS:
        raise ValueError(f"{name}: not an image description")
    dtype, ch = IMAGE_FORMATS[meta["format"]]
    w, h = int(meta["width"]), int(meta["height"])
    path = os.path.join(d, name + ".asset")
    expected = w * h * ch * np.dtype(dtype).itemsize
    if os.path.getsize(path) != expected:
        raise ValueError(f"{path}: file size != {expected} expected from its metadata")
    raw = np.fromfile(path, dtype=np.dtype(dtype).newbyteorder("<")).reshape(h, w, ch)
    return raw, meta["format"]


def srgb_to_rgb(byte_values):
    """core/color.hpp:13-26, in double like the reference."""
    x = np.asarray(byte_values, dtype=np.float64) / 255.0
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def decode_image(raw, fmt):
    """Raw pixels -> linear rgb doubles (height, width, 3) as image::rgb() returns them
    (r8: (v/255, 0, 0), r8_image.hpp; rgb8: v/255; srgb8: srgb_to_rgb; rgb32: as stored)."""
    h, w = raw.shape[:2]
    out = np.zeros((h, w, 3), dtype=np.float64)
    if fmt == "r8":
        out[:, :, 0] = raw[:, :, 0] / 255.0
    elif fmt == "rgb8":
        out[:] = raw / 255.0
    elif fmt == "srgb8":
        out[:] = srgb_to_rgb(raw)
    elif fmt == "rgb32":
        out[:] = raw.astype(np.float64)
    else:
        raise ValueError(fmt)
    return out


def write_room_assets(root, tess=64):
    """Writes the stand-in `walls`, `lights_1`, `lights_2` meshes of scenes.room under <root>/assets."""
    from . import scenes as S

    for name, (pos, idx) in S.room_meshes(tess).items():
        write_mesh(mesh_path(root, name), pos, idx)
