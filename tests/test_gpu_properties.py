"""Size-independent properties of the GPU path at larger sizes, and the C-ABI's error behaviour."""
import ctypes as C

import numpy as np
import pytest

from common import A, R, S, cards_rays, cards_scene, ribbon_scene

pytestmark = pytest.mark.gpu


def render(scene, precision="f32", spp=2, seed=0, tiles=None, cap=0, begin=0, integrator=A.INTEGRATOR_MIS):
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision=precision, max_paths_in_flight=cap)
        ctx.upload_scene(scene)
        ctx.set_sampler(seed)
        for t in (tiles or [None]):
            ctx.render_pass(begin, spp, integrator, 10, True, t)
        rgb, n = ctx.read_film()
        c = ctx.counters()
    assert c.stack_overflows == 0
    return rgb, n, c


def test_deterministic_and_batching_invariant():
    sc = S.room(width=320, height=180, tess=16)
    a, n, _ = render(sc, spp=3)
    b, m, _ = render(sc, spp=3)
    assert n == m and np.array_equal(a, b)  # no float atomics anywhere: bitwise reproducible
    c, _, _ = render(sc, spp=3, cap=4096)   # many small batches (pixel ranges of one sample)
    assert np.array_equal(a, c)
    d, _, _ = render(sc, spp=3, cap=1 << 18)  # several samples per batch
    assert np.array_equal(a, d)


def test_film_does_not_depend_on_the_streaming_grid(monkeypatch):
    """k_generate / k_shade / k_miss / k_accumulate are grid-stride loops (k_shade software-pipelined over its trips: the
    records of the next trip are requested during the current one). With one block per CU the 172 800 paths of this pass
    take three trips of the whole grid (256 CUs) where the default 128 blocks per CU take one: same film, same ray counts,
    bit for bit, in both precisions."""
    sc = S.room(width=320, height=180, tess=16)
    for precision in ("f32", "f64"):
        monkeypatch.delenv("FCGPU_STREAM_GRID_MULT", raising=False)
        a, n, ca = render(sc, precision=precision, spp=3)
        monkeypatch.setenv("FCGPU_STREAM_GRID_MULT", "1")  # read at fcgpu_create
        b, m, cb = render(sc, precision=precision, spp=3)
        assert n == m and np.array_equal(a, b)
        assert (ca.closest_rays, ca.shadow_rays, ca.nan_samples) == (cb.closest_rays, cb.shadow_rays, cb.nan_samples)


def test_tile_sharding_is_bit_exact():
    sc = S.room(width=256, height=144, tess=8)
    full, n, _ = render(sc, spp=2)
    acc = np.zeros_like(full)
    cnt = 0
    for rank in range(4):
        part, k, _ = render(sc, spp=2, tiles=R.shard_tiles(256, 144, rank, 4, 32))
        acc += part
        cnt += k
    assert cnt == n and np.array_equal(acc, full)


def test_sample_pass_split_sums():
    sc = S.room(width=128, height=72, tess=8)
    full, n, _ = render(sc, spp=4, precision="f64")
    a, n1, _ = render(sc, spp=2, begin=0, precision="f64")
    b, n2, _ = render(sc, spp=2, begin=2, precision="f64")
    assert n1 + n2 == n
    assert np.allclose(a + b, full, rtol=1e-13, atol=0)


def test_checkpoint_and_resume_is_bit_exact():
    """fcgpu_save_film / fcgpu_restore_film: a render saved after samples [0, 3), restored into a FRESH context and continued
    with [3, 8) ends with the film of an uninterrupted render of [0, 8), bit for bit, in both precisions; the checkpoint is
    the raw device sums (read_film's rgb_sum divided by W*H would not invert exactly); a checkpoint of another resolution
    is refused by the host wrapper, a null pointer by the library."""
    sc = S.room(width=160, height=90, tess=8)
    for precision in ("f32", "f64"):
        full, n, _ = render(sc, precision=precision, spp=8)
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=precision)
            ctx.upload_scene(sc)
            ctx.set_sampler(0)
            ctx.render_pass(0, 3)
            raw, k = ctx.save_film()
            part, k2 = ctx.read_film()
        assert k == k2 == 160 * 90 * 3
        assert np.allclose(raw * (160.0 * 90.0), part, rtol=1e-15, atol=0)
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=precision)
            ctx.upload_scene(sc)
            ctx.set_sampler(0)
            ctx.restore_film(raw, k)
            ctx.render_pass(3, 5)
            resumed, m = ctx.read_film()
            assert ctx.counters().samples == n
            with pytest.raises(ValueError):
                ctx.restore_film(raw[:10], k)
            assert ctx.lib.fcgpu_restore_film(ctx.h, None, 0) == A.E_INVALID
            assert ctx.lib.fcgpu_save_film(ctx.h, None, None) == A.E_INVALID
        assert m == n and np.array_equal(resumed, full)


def test_emission_linearity():
    """Scaling every emitter by 2 scales the image by exactly 2 (power of two: exact in fp)."""
    sc1 = S.room(width=96, height=54, tess=4)
    sc2 = S.room(width=96, height=54, tess=4)
    for e in sc2.entities:
        e.emission_strength *= 2.0
    a, _, _ = render(sc1, spp=2)
    b, _, _ = render(sc2, spp=2)
    assert np.array_equal(2.0 * a, b)


def test_tessellation_invariance():
    """Flat tessellation of the walls leaves direct lighting sample-identical (max_path_length 2) and
    the full image statistically unchanged (longer paths differ per sample: the tangent frame of a
    mesh hit follows its triangle's first edge, mesh_surface.hpp:221-248)."""
    def go(tess, mpl, spp):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision="f64")
            ctx.upload_scene(S.room(width=96, height=54, tess=tess))
            ctx.render_pass(0, spp, A.INTEGRATOR_MIS, mpl, True)
            return ctx.read_film()[0]

    a, b = go(2, 2, 4), go(32, 2, 4)
    rel = np.abs(a - b).max(axis=2) / np.maximum(a.max(axis=2), 1e-3 * a.mean())
    assert (rel <= 1e-6).mean() >= 0.97  # vertices of finer grids are rounded to float at other places
    a, b = go(2, 10, 64), go(32, 10, 64)
    assert abs(a.mean() - b.mean()) <= 0.02 * a.mean()


def test_bsdf_and_mis_agree_in_expectation():
    sc = S.room(width=64, height=36, tess=4)
    a, n, _ = render(sc, spp=256, integrator=A.INTEGRATOR_MIS)
    b, m, _ = render(sc, spp=256, integrator=A.INTEGRATOR_BSDF)
    assert abs(a.mean() / n - b.mean() / m) <= 0.05 * (a.mean() / n)


def test_full_size_pass_counters():
    """One sample of the headline frame: every path traced, ray budget as the survey measured."""
    sc = S.room(width=1920, height=1080, tess=64)
    rgb, n, c = render(sc, spp=1)
    assert n == 1920 * 1080
    rays = (c.closest_rays + c.shadow_rays) / n
    assert 17.5 <= rays <= 19.5  # 18.47 rays / sample
    assert np.isfinite(rgb).all() and rgb.min() >= 0 and c.nan_samples == 0


def test_error_codes():
    lib = R.load_library()
    ctx = R.GpuContext(0)
    with pytest.raises(R.FcgpuError) as e:
        ctx.render_pass(0, 1)
    assert e.value.code == A.E_NOSCENE
    sc = S.room_planes(32, 18)
    ctx.upload_scene(sc)
    with pytest.raises(R.FcgpuError) as e:
        ctx.render_pass(0, 1, tile=(0, 0, 64, 64))
    assert e.value.code == A.E_INVALID
    with pytest.raises(R.FcgpuError) as e:
        ctx.render_pass(0, 1, integrator=7)
    assert e.value.code == A.E_INVALID
    bad = S.room_planes(32, 18)
    bad.entities[0].material = 99
    with pytest.raises(R.FcgpuError) as e:
        ctx.upload_scene(bad)
    assert e.value.code == A.E_INVALID
    assert b"entity" in lib.fcgpu_last_error(ctx.h)
    # texture environment light: image index and distribution resolution are validated at upload
    for field, value in (("image", 5), ("distribution_width", 0), ("integral_sample_count", 0), ("filter", 9)):
        bad = S.env_texture_scene(16, 12)
        setattr(bad.env_texture, field, value)
        with pytest.raises(R.FcgpuError) as e:
            ctx.upload_scene(bad)
        assert e.value.code == A.E_INVALID
    # image textures of materials: handle range, parameters (incl. normal maps) the material class does not have
    for index, field, value in ((0, "tex_color_a", 99), (0, "tex_color_a", -1), (0, "tex_roughness", 1), (5, "tex_color_b", 1),
                                (6, "tex_color_a", 1), (4, "tex_normal", 1), (1, "tex_normal", 2)):  # glass / plastic: no normal_mapping<T>
        bad = S.textured_scene(16, 12)
        setattr(bad.materials[index], field, value)
        with pytest.raises(R.FcgpuError) as e:
            ctx.upload_scene(bad)
        assert e.value.code == A.E_INVALID, (index, field)
        assert b"material" in lib.fcgpu_last_error(ctx.h)
    bad = S.textured_scene(16, 12)
    bad.materials[2].tex_metalness = bad.checker((1, 1, 1), (0, 0, 0), 4.0)  # checker_texture_2d_rgb on a texture_2d_r parameter
    with pytest.raises(R.FcgpuError) as e:
        ctx.upload_scene(bad)
    assert e.value.code == A.E_INVALID
    bad = S.textured_scene(16, 12)
    bad.textures[0].kind = 7
    with pytest.raises(R.FcgpuError) as e:
        ctx.upload_scene(bad)
    assert e.value.code == A.E_INVALID
    bad = S.textured_scene(16, 12)
    bad.textures[0].image = 77
    with pytest.raises(R.FcgpuError) as e:
        ctx.upload_scene(bad)
    assert e.value.code == A.E_INVALID
    with pytest.raises(R.FcgpuError) as e:
        ctx.set_sampler(0, stratified=0)
    assert e.value.code == A.E_INVALID
    for field, value in (("bvh_builder", 3), ("precision", 9), ("max_paths_in_flight", 17)):
        o = A.Options(precision=A.PRECISION_F32, sort_materials=1, epsilon=0.0, max_paths_in_flight=0, bvh_builder=A.BVH_HOST_SAH)
        setattr(o, field, value)
        assert lib.fcgpu_set_options(ctx.h, C.byref(o)) == A.E_INVALID, field
    o = A.Options(precision=A.PRECISION_F32, sort_materials=0, epsilon=0.0, max_paths_in_flight=0, bvh_builder=A.BVH_HOST_SAH)
    assert lib.fcgpu_set_options(ctx.h, C.byref(o)) == A.E_UNSUPPORTED  # documented as "single queue", never built: refused
    ctx.upload_scene(sc)
    with pytest.raises(R.FcgpuError) as e:
        ctx.export_image("/tmp/fcgpu_nothing_rendered")  # 0 / 0 would be an image of NaNs
    assert e.value.code == A.E_INVALID
    for args in ((0, 0, 1), (4, 2, 2), (4, 0, 0)):
        with pytest.raises(R.FcgpuError) as e:
            ctx.render_bands(*args, 0, 1)
        assert e.value.code == A.E_INVALID
    h = C.c_void_p()
    assert lib.fcgpu_create(4096, C.byref(h)) == A.E_INVALID
    ctx.close()


def test_texture_environment_estimators_agree_and_importance_sampling_helps():
    """Open scene under texture_infinity_area_light: the BSDF-only and the MIS integrator converge to
    the same image, and at equal samples MIS (which importance-samples the sun) is the less noisy one."""
    sc = S.env_texture_scene(48, 32, lens=0.0)
    ref = render(sc, spp=512, seed=77)
    ref = ref[0] / ref[1]
    out = {}
    for name, integ in (("mis", A.INTEGRATOR_MIS), ("bsdf", A.INTEGRATOR_BSDF)):
        rgb, n, _ = render(sc, spp=256, seed=3, integrator=integ)
        out[name] = rgb / n
    assert abs(out["mis"].mean() - out["bsdf"].mean()) < 0.05 * ref.mean()
    mse = {k: float(np.mean((np.clip(v, 0, 10) - np.clip(ref, 0, 10)) ** 2)) for k, v in out.items()}
    print("MSE vs 512-spp MIS reference:", mse)
    assert mse["mis"] < mse["bsdf"]


def test_empty_and_degenerate_inputs():
    sc = S.room_planes(16, 9)
    with R.GpuContext(0) as ctx:
        ctx.upload_scene(sc)
        assert len(ctx.trace_closest(np.zeros(0, dtype=A.ray_dtype))) == 0
        ctx.render_pass(0, 0)  # zero samples: a no-op
        rgb, n = ctx.read_film()
        assert n == 0 and rgb.max() == 0
        rays = np.zeros(3, dtype=A.ray_dtype)
        rays["origin"] = [[0, 4, 5], [0, 4, 5], [0, 100, 5]]
        rays["direction"] = [[0, 1, 0], [0, 0, 1], [0, 1, 0]]  # axis-parallel rays
        rays["t_max"] = [np.inf, 1.0, np.inf]
        hits = ctx.trace_closest(rays)
        assert hits["prim"][0] == 6 and abs(hits["t"][0] - 3.9) < 1e-5  # the plane emitter at y = 7.9
        assert hits["prim"][1] == 0xFFFFFFFF  # t_max shorter than the wall distance
        assert hits["prim"][2] == 0xFFFFFFFF  # outside the room, pointing away


def test_cpp_host_renders_the_room_from_asset_files(tmp_path):
    """csrc/demo_room (C++ host over the C-ABI, meshes read by fc_assets.hpp from the reference's
    .mesh format) against the Python host on the generated scene: same library, same scene."""
    import importlib
    import os
    import subprocess

    from common import ROOT, pixel_rel_err

    AS = importlib.import_module("ray-tracing_amd.assets")
    exe = os.path.join(ROOT, "ray-tracing_amd", "csrc", "build", "demo_room")
    assert os.path.exists(exe), "build csrc/build/demo_room first (make -C ray-tracing_amd/csrc)"
    AS.write_room_assets(str(tmp_path), tess=8)
    w, h, spp = 96, 54, 4
    out = subprocess.run([exe, str(w), str(h), str(spp), str(tmp_path)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    img = np.fromfile(tmp_path / "room.raw", dtype=np.float32).reshape(h, w, 3)
    rgb, n, _ = render(S.room(w, h, tess=8, assets_root=str(tmp_path)), spp=spp)
    ref = (rgb / n).reshape(h, w, 3)
    # the two hosts build the camera angles with differently rounded degree->radian products, so a
    # few paths may flip a discrete decision: compare like the fp32 parity tests do
    err = pixel_rel_err(img.astype(np.float64), ref)
    assert np.mean(err < 1e-3) > 0.97 and abs(img.mean() - ref.mean()) < 0.01 * ref.mean()
    # the same host with three contexts (fcgpu::multi_gpu_renderer: bands of rows per context, fcgpu_reduce_film): every pixel
    # has one owner, so the image file is identical byte for byte and the same rays were cast
    out3 = subprocess.run([exe, str(w), str(h), str(spp), str(tmp_path), "3"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out3.returncode == 0 and "3 contexts" in out3.stdout, out3.stdout + out3.stderr
    img3 = np.fromfile(tmp_path / "room.raw", dtype=np.float32).reshape(h, w, 3)
    assert np.array_equal(img, img3)
    assert out.stdout.split("rays")[1].split()[0] == out3.stdout.split("rays")[1].split()[0]
    # the same host stopped after 1 of its 4 samples per pixel, checkpointed to a file (gpu_renderer::save_checkpoint) and
    # finished by a NEW renderer that loads it: the image an uninterrupted run writes, byte for byte
    outc = subprocess.run([exe, str(w), str(h), str(spp), str(tmp_path), "1", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert outc.returncode == 0, outc.stdout + outc.stderr
    assert os.path.getsize(tmp_path / "room.ckpt") == 32 + 24 * w * h
    imgc = np.fromfile(tmp_path / "room.raw", dtype=np.float32).reshape(h, w, 3)
    assert np.array_equal(img, imgc)


def test_deep_bvh_spills_the_traversal_stack_and_still_matches_the_oracle():
    """A mesh whose SAH tree is very unbalanced (triangle sizes in geometric progression) needs more
    traversal-stack entries than the LDS holds: the rest spills to global memory (fc_traverse.h)."""
    from oracle import pyref

    n, g = 420, 1.1
    x0 = g ** np.arange(n + 1) - 1.0             # strip of quads with exponentially growing width (up to ~2e17)
    pos = np.zeros((2 * (n + 1), 3), np.float32)
    pos[0::2, 0], pos[1::2, 0] = x0, x0
    pos[1::2, 2] = 1.0
    pos[:, 1] = 0.25 * pos[:, 0] * np.where(np.arange(2 * (n + 1)) % 2 == 0, 1.0, -1.0)  # a twisted ribbon whose height grows with x: every box of the chain straddles y = 0
    k = np.arange(n, dtype=np.uint32) * 2
    idx = np.stack([k, k + 1, k + 2, k + 2, k + 1, k + 3], 1).reshape(-1)
    sc = S.Scene()
    sc.entity(sc.mesh(pos, idx), sc.diffuse((0.7, 0.7, 0.7)))
    sc.set_env((1.0, 0.9, 0.8))  # no analytic primitive here: their quadratics are noise at coordinates of 1e30
    sc.set_camera((20.0, 6.0, 6.0), (np.radians(40.0), np.radians(180.0), 0.0), np.radians(60.0), 48, 32)
    rng = np.random.default_rng(5)
    rays = np.zeros(20000, dtype=A.ray_dtype)
    ox = g ** rng.uniform(0, n, len(rays))       # log-uniform along the strip, so every scale of the tree is visited
    rays["origin"] = np.stack([ox, 0.3 * ox + rng.uniform(0.5, 1.0, len(rays)), rng.uniform(-0.5, 1.5, len(rays))], 1)
    d = rng.standard_normal((len(rays), 3))
    d[:, 0] *= 0.2
    d[:, 1] = -np.abs(d[:, 1]) - 0.2
    # a quarter of the rays run ALONG the ribbon from its big end towards the small one, inside every
    # box of the chain: the SAH build peels the small triangles off first, so the near child is the
    # interior remainder and the far (leaf) children pile up on the stack level after level
    m = len(rays) // 4
    slope = rng.uniform(-0.2, 0.2, m)
    X = g ** rng.uniform(300, 400, m)
    # (+3: the line misses the smallest triangles by a clear margin instead of grazing them, where
    # double precision from 1e15 away decides hit or miss by rounding)
    rays["origin"][:m] = np.stack([X, slope * X + 3.0, rng.uniform(0.1, 0.9, m)], 1)
    d[:m] = np.stack([-np.ones(m), -slope, np.zeros(m)], 1)  # dz = 0: the ribbon is only 1 wide in z
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["t_max"] = np.inf
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f64")
        ctx.upload_scene(sc)
        c = ctx.counters()
        assert c.reserved[1] > 30, f"scene is not deep enough to exercise the spill path (max_stack {c.reserved[1]})"
        hits = ctx.trace_closest(rays)
        occ = ctx.trace_any(rays)
        ctx.render_pass(0, 4)
        rgb, cnt = ctx.read_film()
        c = ctx.counters()
        assert c.stack_overflows == 0
        assert c.reserved[5] > 0, "no push reached the spill area: the test does not cover it"
    o = pyref.restatement("f64").scene(sc)
    ref = o.trace_closest(rays)
    # Rays that start 1e13..1e16 away see the small end of the ribbon collapse onto a few points in
    # double precision: several triangles then report the SAME distance and the winner depends on the
    # visiting order (binary tree in the reference, 4-wide here). Distances must agree everywhere; ids
    # wherever the hit is not such a tie.
    hit = ref["prim"] != 0xFFFFFFFF
    hm = np.flatnonzero((hits["prim"] != 0xFFFFFFFF) != hit)
    print("hit/miss disagreements:", len(hm), "in the along-ribbon quarter:", int((hm < len(rays) // 4).sum()),
          [(int(i), int(hits["prim"][i]), int(ref["prim"][i]), float(hits["t"][i]), float(ref["t"][i])) for i in hm[:8]])
    assert np.array_equal(hits["prim"] != 0xFFFFFFFF, hit)
    assert np.allclose(hits["t"][hit], ref["t"][hit], rtol=1e-9, atol=0)
    bad = hits["prim"] != ref["prim"]
    print("tie mismatches among the along-ribbon rays:", int(bad.sum()))
    assert bad.mean() < 0.01 and not bad[len(rays) // 4:].any()  # none among the ordinary rays
    assert np.array_equal(occ.astype(bool), ref["prim"] != 0xFFFFFFFF)
    film = o.render(4)
    assert pixel_rel_err_max(rgb, film[0]) <= 1e-9


@pytest.mark.parametrize("n,depth", [(58, 28), (60, 29), (66, 30), (67, 31), (72, 32)])
def test_bvh_whose_stack_need_is_at_the_lds_capacity(n, depth):
    """Worst-case stack needs of 28 (the deepest the kernels without spill code take), 29 and 30 (would fit the 30 LDS
    entries, but those kernels stop pushing once 27 are occupied: such scenes must get the SPILL instantiation), 31
    and 32 -- on a scene whose rays really fill the stack to that worst case (common.cards_scene): no push is lost,
    hits equal the oracle's in fp64 and, up to edge ties, in fp32."""
    from oracle import pyref

    sc = cards_scene(n)
    rays = cards_rays(6000, n)
    ref = pyref.restatement("f64").scene(sc).trace_closest(rays)
    assert (ref["prim"] >= n).mean() > 0.5 and (ref["prim"] < n).mean() > 0.1  # backstop after a full stack / card hits
    for precision in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=precision)
            ctx.upload_scene(sc)
            assert ctx.counters().reserved[1] == depth
            hits = ctx.trace_closest(rays)
            occ = ctx.trace_any(rays)
            ctx.render_pass(0, 2)
            c = ctx.counters()
        assert c.stack_overflows == 0
        assert (int(c.reserved[5]) > 0) == (depth > 30)  # pushes beyond the LDS entries happen exactly when the tree needs them
        mismatch = hits["prim"] != ref["prim"]
        assert mismatch.sum() == 0 if precision == "f64" else mismatch.mean() < 0.01
        assert np.allclose(hits["t"][~mismatch], ref["t"][~mismatch], rtol=1e-9 if precision == "f64" else 1e-5, atol=0)
        assert occ.all()


@pytest.mark.parametrize("n,depth", [(50, 25), (58, 28), (72, 32)])
def test_occlusion_rays_of_the_fp32_path_spill_beyond_their_shorter_lds_stack(n, depth):
    """The fp32 shadow kernel keeps 22 entries per lane in LDS, not 30 (fc_types.h: SHADOW_STACK_CAP: seven waves per
    SIMD), and fcgpu_trace_any runs its traversal: on hierarchies whose worst case lies between the two capacities (25, 28) only
    the occlusion rays use the global spill columns, beyond 30 both kinds do; every answer equals the fp64 device's and the
    CPU restatement's, no push is lost."""
    from oracle import pyref

    sc = cards_scene(n, mirror=True)  # the interior child in the first slot: rays that visit in slot order fill the stack too
    rays = cards_rays(6000, n, mirror=True)
    # besides the rays that pierce every box: rays that end between two cards (unoccluded) and rays that hit one
    rays["t_max"][1::3] = np.random.default_rng(n).uniform(1.0, 1.2 ** (n - 1), len(rays[1::3]))
    o = pyref.restatement("f64").scene(sc)
    ref = o.trace_any(rays).astype(bool)
    assert 0.2 < ref.mean() < 1.0
    got = {}
    for precision in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=precision)
            ctx.upload_scene(sc)
            assert ctx.counters().reserved[1] == depth
            ctx.reset_counters()
            got[precision] = ctx.trace_any(rays).astype(bool)
            c = ctx.counters()
            assert c.stack_overflows == 0
            # fp64: 30 entries in LDS; fp32 occlusion rays: 22
            assert (int(c.reserved[5]) > 0) == (depth > (30 if precision == "f64" else 22)), (precision, int(c.reserved[5]))
    assert np.array_equal(got["f64"], ref)
    # fp32 decides grazing card edges and ray ends by rounding: a handful of rays may differ
    assert (got["f32"] != ref).mean() < 2e-3


def pixel_rel_err_max(a, b):
    from common import pixel_rel_err
    return float(pixel_rel_err(a, b).max())


def test_textured_material_with_uniform_images_equals_its_constant_twin():
    """Image textures whose texels all hold one value must render exactly what the constant material
    renders (same lobes, same sample stream): per-hit lobe construction == upload-time precompute, and a flat
    normal map changes nothing.
    All parameter values are dyadic so that fp32 and fp64 parameter arithmetic agree to the bit."""
    def build(textured):
        sc = S.Scene()
        flat = lambda c: np.broadcast_to(np.asarray(c, np.float32), (3, 5, 3)).copy()
        tex = (lambda c: sc.texture(sc.image(flat(c), "rgb32"), "box")) if textured else None  # box: texel values come back exactly
        def mat(kind, **kw):
            names = {"color_a": "tex_color_a", "color_b": "tex_color_b", "ior": "tex_ior", "metalness": "tex_metalness"}
            args, texargs = dict(kw), {}
            if textured:
                for k, v in kw.items():
                    if k in names and not (k == "ior" and kind not in (A.MATERIAL_PLASTIC, A.MATERIAL_STANDARD)):
                        texargs[names[k]] = tex(v if k.startswith("color") else (v, 0, 0))
                if kind == A.MATERIAL_STANDARD:
                    texargs["tex_roughness"] = tex((kw["roughness"][0], 0, 0))
                if kind in (A.MATERIAL_STANDARD, A.MATERIAL_DIFFUSE):
                    texargs["tex_normal"] = tex((0.5, 0.5, 1.0))  # a flat normal map: normal_mapping's skip path
            return sc.material(kind, **args, **texargs)
        sc.entity(sc.plane((0, 0, 0), (16, 16)), mat(A.MATERIAL_DIFFUSE, color_a=(0.25, 0.5, 0.75)))
        sc.entity(sc.plane((0, 4, -6), (16, 8), (np.pi / 2, 0, 0)),
                  mat(A.MATERIAL_PLASTIC, color_a=(0.5, 0.25, 0.125), color_b=(1, 1, 1), roughness=(0.25, 0.125), ior=1.5))
        sc.entity(sc.sphere((-2.5, 1.2, 0.5), 1.2), mat(A.MATERIAL_STANDARD, color_a=(0.75, 0.5, 0.25), roughness=(0.25, 0.25), ior=1.5, metalness=0.5))
        sc.entity(sc.sphere((2.5, 1.0, 1.0), 1.0), mat(A.MATERIAL_STANDARD, color_a=(0.75, 0.75, 0.25), roughness=(0.0, 0.0), ior=1.25, metalness=1.0))
        sc.entity(sc.sphere((0.2, 0.9, 2.2), 0.9), mat(A.MATERIAL_GLASS, color_a=(1, 1, 1), color_b=(0.5, 1, 0.75), roughness=(0.0, 0.0)),
                  medium=sc.medium(1, 1.5, (0.9, 0.6, 0.4), 0.3))
        sc.entity(sc.sphere((0.0, 5.0, 0.0), 0.7), sc.diffuse((0.1, 0.1, 0.1)), emission=(1, 1, 1), strength=15.0)
        sc.set_env((0.3, 0.35, 0.45))
        sc.set_camera((0.0, 3.2, 9.5), (np.radians(12.0), np.radians(180.0), 0.0), np.radians(42.0), 64, 40)
        return sc
    for precision in ("f64", "f32"):
        a, na, _ = render(build(False), spp=8, seed=5, precision=precision)
        b, nb, _ = render(build(True), spp=8, seed=5, precision=precision)
        assert na == nb and a.mean() > 0
        if precision == "f64":
            assert np.array_equal(a, b)
        else:
            # the two run different kernel instantiations (TEX on / off): fp32 contraction may differ by an ulp, and a
            # path whose discrete decision flips on that ulp diverges; same bounds as the fp32 parity test
            rel = np.abs(a - b).max(2) / np.maximum(np.abs(a).max(2), 1e-3 * a.mean())
            assert np.median(rel) <= 1e-4 and (rel <= 1e-2).mean() >= 0.90 and abs(a.mean() - b.mean()) <= 0.02 * a.mean()


def test_texture_coordinates_outside_the_unit_square_are_clamped():
    """Mesh uvs in [-0.6, 1.7]: both filters clamp to the edge texels (the reference indexes out of bounds below 0 with
    its box filter; device and restatement clamp), checked against the restatement in fp64."""
    from oracle import pyref

    sc = S.Scene()
    tex_box = sc.texture(sc.image(S.pattern_image(8, 8, "rgb8", 3), "rgb8"), "box")
    tex_lin = sc.texture(sc.image(S.pattern_image(6, 5, "rgb32", 4, 0.1, 0.9), "rgb32"), "bilinear")
    for k, (tex, x) in enumerate(((tex_box, -2.2), (tex_lin, 2.2))):
        qp, qi = S.grid_quad((x - 2.0, 0.0, 2.0), (4.0, 0, 0), (0, 0, -4.0), 3, 3)
        uv = (qp[:, [0, 2]] - qp[:, [0, 2]].min(0)) / 4.0 * 2.3 - 0.6
        sc.entity(sc.mesh(qp, qi, uvs=uv), sc.material(A.MATERIAL_DIFFUSE, tex_color_a=tex))
    sc.entity(sc.sphere((0.0, 4.0, 0.0), 0.5), sc.diffuse((0.1, 0.1, 0.1)), emission=(1, 1, 1), strength=30.0)
    sc.set_env((0.4, 0.4, 0.4))
    sc.set_camera((0.0, 6.0, 6.0), (np.radians(45.0), np.radians(180.0), 0.0), np.radians(50.0), 64, 40)
    ref, n, _ = pyref.restatement("f64").scene(sc).render(4, seed=3, threads=4)
    rgb, m, c = render(sc, precision="f64", spp=4, seed=3)
    assert n == m and c.nan_samples == 0 and ref.mean() > 0
    assert (np.abs(rgb - ref).max(2) / np.maximum(np.abs(ref).max(2), 1e-3 * ref.mean())).max() <= 1e-9


def test_stratified_renderer_runs_the_samplers_sample_count():
    """stratified_sampler{10, seed}.get_sample_count() is floor(sqrt(10))^2 = 9 (stratified_sampler.hpp:15-16, 26-29) and
    renderer::run_pixel renders that many samples per pixel (renderer.hpp:159): so does the drop-in renderer."""
    sc = S.room_planes(32, 18)
    r = R.GpuRenderer(sc, sample_count=10, sampler="stratified", seed=3)
    r.run()
    _, n = r.film()
    r.close()
    assert n == 9 * 32 * 18
    r = R.GpuRenderer(sc, sample_count=10, sampler="random", seed=3)
    r.run()
    _, n = r.film()
    r.close()
    assert n == 10 * 32 * 18


def test_band_sharding_is_bit_exact_and_reduces_in_the_library():
    """fcgpu_render_bands: every rank renders its bands of rows as ONE pass; fcgpu_reduce_film sums the per-context films
    on the device (here: four contexts on the one GPU of the box). The result equals the single-context film bit for
    bit, for band heights that do and do not divide the frame, and more ranks than bands."""
    sc = S.room(width=200, height=75, tess=8)
    full, n, cf = render(sc, spp=3)
    for band_rows, world in ((4, 4), (16, 3), (64, 4)):
        ctxs = []
        for rank in range(world):
            ctx = R.GpuContext(0)
            ctx.set_options(precision="f32")
            ctx.upload_scene(sc)
            ctx.set_sampler(0)
            ctx.render_bands(band_rows, rank, world, 0, 3)
            ctxs.append(ctx)
        R.reduce_film(ctxs, root=1)
        rgb, cnt = ctxs[1].read_film()
        rays = sum(int(c.counters().closest_rays) for c in ctxs)
        copied = int(ctxs[1].counters().reduce_bytes_copied)
        others = [c.read_film() for i, c in enumerate(ctxs) if i != 1]
        for c in ctxs:
            c.close()
        assert cnt == n and np.array_equal(rgb, full), (band_rows, world)
        assert rays == int(cf.closest_rays)
        assert all(k == 0 and a.max() == 0 for a, k in others)  # contributors start over after a reduce
        # a contributor that rendered bands sends its bands, not the frame: the copies moved exactly the rows the other ranks own
        rows_of = lambda rank: sum(min(band_rows, 75 - y0) for y0 in range(rank * band_rows, 75, world * band_rows))
        assert copied == sum(rows_of(k) for k in range(world) if k != 1) * 200 * 24, (band_rows, world, copied)


def test_reduce_film_sums_sample_passes():
    """Sample-pass sharding through the C-ABI reduce: two contexts render disjoint sample ranges of the full frame; the
    reduced film equals the single-context film up to the order of the fp64 additions."""
    sc = S.room(width=96, height=54, tess=4)
    full, n, _ = render(sc, spp=4, precision="f64")
    ctxs = []
    for rank in range(2):
        ctx = R.GpuContext(0)
        ctx.set_options(precision="f64")
        ctx.upload_scene(sc)
        ctx.set_sampler(0)
        ctx.render_pass(2 * rank, 2)
        ctxs.append(ctx)
    R.reduce_film(ctxs, root=0)
    rgb, cnt = ctxs[0].read_film()
    assert int(ctxs[0].counters().reduce_bytes_copied) == 96 * 54 * 24  # a contributor that rendered the frame sends the frame
    with pytest.raises(R.FcgpuError) as e:
        R.reduce_film([ctxs[0], ctxs[0]], root=0)
    assert e.value.code == A.E_INVALID
    for c in ctxs:
        c.close()
    assert cnt == n and np.allclose(rgb, full, rtol=1e-13, atol=0)


def test_decision_log_records_what_the_counters_count():
    """fcgpu_set_decision_log (parity hook): one hit word and one flag word per vertex and path. Its totals must equal
    the counters of the same pass (shadow rays cast / occluded, paths that left the scene), it must not change the
    film, and a second render logs the same words."""
    sc = S.nested_media(32, 32)
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32")
        ctx.upload_scene(sc)
        ctx.set_sampler(4)
        ctx.render_pass(0, 3)
        plain, _ = ctx.read_film()
        ctx.clear_film()
        ctx.reset_counters()
        ctx.set_decision_log(True)
        ctx.render_pass(0, 3)
        hits, flags = ctx.read_decision_log()
        logged, _ = ctx.read_film()
        c = ctx.counters()
        ctx.clear_film()
        ctx.render_pass(0, 3)
        hits2, flags2 = ctx.read_decision_log()
        with pytest.raises(R.FcgpuError) as e:  # the log needs the pass in one batch
            ctx.set_options(precision="f32", max_paths_in_flight=256)
            ctx.render_pass(0, 3)
        assert e.value.code == A.E_LIMIT
    assert hits.shape == (10, 32 * 32 * 3)
    assert np.array_equal(plain, logged)
    assert np.array_equal(hits, hits2) and np.array_equal(flags, flags2)
    assert int(((flags & 8) != 0).sum()) == int(c.shadow_rays)
    assert int(((flags & 16) != 0).sum()) == int(c.shadow_occluded)
    assert ((flags & 16) == 0)[(flags & 8) == 0].all()          # only a cast ray can be occluded
    shaded = (flags & 1) != 0
    assert (hits[shaded] < 0xFFFFFFFD).all()                    # what was shaded had been hit
    assert (hits[0] != 0xFFFFFFFF).all()                        # every path has a first vertex (hit, miss or dead)
    went_on = (flags[:-1] & 2) != 0
    assert (hits[1:][went_on] != 0xFFFFFFFF).all() and (hits[1:][~went_on] == 0xFFFFFFFF).all()


def test_batch_size_is_halved_when_the_device_cannot_hold_the_queues():
    """With the GPU nearly full (ballast allocated through torch) a pass whose queues do not fit must not fail: the
    library halves the batch size until they do (fcgpu_options.max_paths_in_flight, FCGPU_E_NOMEM only below 2^16 paths)
    and renders the same film, bit for bit, in more and smaller batches."""
    import torch

    sc = S.room(width=1920, height=1080, tess=8)
    want, n, _ = render(sc, spp=4, cap=1 << 20)  # the reference film, rendered in 1 M-path batches
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32")
        ctx.upload_scene(sc)
        ctx.set_sampler(0)
        torch.cuda.synchronize()
        free, _total = torch.cuda.mem_get_info(0)
        keep = 1 << 30  # leave 1 GiB: the 8.3 M-path pass asks for ~2 GiB of queues
        ballast = torch.empty(max(0, free - keep), dtype=torch.uint8, device="cuda:0")
        try:
            ctx.render_pass(0, 4)
            rgb, m = ctx.read_film()
            c = ctx.counters()
        finally:
            del ballast
            torch.cuda.empty_cache()
    assert m == n and c.stack_overflows == 0
    assert c.kernel_launches >= 2 * 69  # at least two batches' worth of launches (69 each): the pass was split
    assert np.array_equal(rgb, want)


def test_8k_frame_dealt_in_bands_to_eight_ranks_is_bit_exact():
    """BASELINE config 5's one-GPU half (7680 x 4320, tile-sharded over 8 GPUs): the 8 ranks' shares of the 8K frame, rendered
    through fcgpu_render_bands as 8 calls (bands of 4 rows dealt round-robin, one pass per rank), together cover every pixel
    once, and bands owned by three different ranks equal a plain fcgpu_render_pass of the same rows bit for bit."""
    W, H, rows, world = 7680, 4320, 4, 8
    sc = S.room(width=W, height=H, tess=64)
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32")
        ctx.upload_scene(sc)
        ctx.set_sampler(0)
        for rank in range(world):
            ctx.render_bands(rows, rank, world, 0, 1)
        gathered, n = ctx.read_film()
        c = ctx.counters()
        assert n == W * H and c.stack_overflows == 0 and c.nan_samples == 0
        nbands = H // rows
        picks = [1, 8 * (nbands // 16) + 3, nbands - 2]  # owners 1, 3 and 6
        assert sorted({b % world for b in picks}) == [1, 3, 6]
        kept = {b: gathered[b * rows:(b + 1) * rows].copy() for b in picks}
        lit = float((gathered.max(axis=2) > 0).mean())
        del gathered
        ctx.clear_film()
        for b in picks:
            ctx.render_pass(0, 1, tile=(0, b * rows, W, rows))
        plain, m = ctx.read_film()
    assert m == len(picks) * rows * W
    assert lit > 0.9  # a closed room: nearly every pixel of every rank's share received light
    for b in picks:
        assert np.array_equal(plain[b * rows:(b + 1) * rows], kept[b]), b


def test_paths_nested_nine_media_deep_use_the_extension_word():
    """The priority-medium stack holds the reference's 9 media (integrator.hpp:24): entries 6..9 live in the extension word
    beside the path state. nested_deep(9) must really reach them (medium_deep_pushes > 0), lose nothing, and -- the
    parity itself is test_gpu_parity's film_nested_deep9* against the real reference -- render the same film in small batches."""
    sc = S.nested_deep(40, 40, depth=9)
    out = []
    for cap in (0, 512):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision="f64", max_paths_in_flight=cap)
            ctx.upload_scene(sc)
            ctx.set_sampler(7)
            ctx.render_pass(0, 4, A.INTEGRATOR_MIS, 24, True)
            rgb, n = ctx.read_film()
            c = ctx.counters()
        assert c.medium_overflows == 0 and c.reserved[0] == 0
        assert c.medium_deep_pushes > 0, "no path nested deeper than five media: the scene does not cover the extension word"
        out.append((rgb, int(c.medium_deep_pushes), int(c.closest_rays)))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:]


def test_a_tenth_nested_medium_is_refused_not_rendered():
    """Ten nested media are beyond the reference's own capacity (it writes past helper::buffer_). The device drops the push and
    counts it; the counter is an error: GpuRenderer.run() (like gpu_renderer::verify() in C++ and bench.py) refuses the frame."""
    sc = S.nested_deep(40, 40, depth=10)
    r = R.GpuRenderer(sc, integrator="mis", max_path_length=24, sample_count=4, seed=7, precision="f32")
    with pytest.raises(R.FcgpuError) as e:
        r.run()
    assert e.value.code == A.E_LIMIT and "medium_overflows" in str(e.value)
    assert r.ctx.counters().medium_overflows > 0
    r.close()
