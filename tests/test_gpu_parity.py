"""GPU parity tests proper: the HIP path (through the C-ABI) against the golden vectors produced
by the real reference and against the CPU oracle on identical sample streams.

Bars (stated here, repeated in DESIGN.md):
  * fp64 mode: per-pixel radiance within 1e-9 relative of the reference (same algorithm, same
    precision; only libm ulps and FMA contraction differ); closest-hit primitive ids identical on
    the dumped ray sets; ray counts equal up to 1e-3 relative.
  * fp32 mode (the fast path): closest-hit ids identical except a near-tie fraction <= 1e-3, every
    differing id explained by distance (classify_id_mismatches), |t_gpu - t_ref| <= 1e-4 * max(1, t_ref);
    per-pixel radiance on identical streams at the fixtures' 2..20 spp: median relative error <= 5e-6 and
    >= 99 % of pixels within 1e-2 (a path whose discrete decision flips on a rounding legitimately
    diverges; those pixels are not bounded), image mean within 0.5 %. At the headline configuration
    (1080p, tess 64, 16 spp) the same bar is tests/test_gpu_headline.py's.
"""
import ctypes as C
import os

import numpy as np
import pytest

from common import A, FILM_FIXTURES, GOLDEN, R, RAY_FIXTURES, S, film_stratified, load_film, load_rays, pixel_rel_err
from oracle import pyref

pytestmark = pytest.mark.gpu


def make_ctx(scene, precision, **opt):
    ctx = R.GpuContext(0)
    ctx.set_options(precision=precision, **opt)
    ctx.upload_scene(scene)
    return ctx


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_closest_hit_ids_f64(name):
    z, scene = load_rays(name)
    with make_ctx(scene, "f64") as ctx:
        hits = ctx.trace_closest(z["rays"])
        anyh = ctx.trace_any(z["rays"])
        assert ctx.counters().stack_overflows == 0
    ref_prim = z["bvh_prim"]
    mismatch = hits["prim"] != ref_prim
    assert mismatch.mean() <= 1e-4, f"prim id mismatches {mismatch.sum()} / {len(ref_prim)}"
    ok = (~mismatch) & (ref_prim != 0xFFFFFFFF)
    assert np.allclose(hits["t"][ok], z["bvh_t"][ok], rtol=1e-9, atol=1e-9)
    assert (anyh != z["any_hit"]).mean() <= 1e-4


def classify_id_mismatches(h, ref_prim, ref_t, ray_t_max):
    """Every closest-hit id that differs from the reference's must be explained by distance: a NEAR-TIE (the two
    primitives are hit at the same distance within 1e-4 * max(1, t): shared edges, coplanar faces), or, for rays whose
    ends sit closer to a surface than fp32 resolves (the reference's own rays: 1e-6), an END-SURFACE case: one side
    reports the surface the ray starts on (t <= 1e-4) or the one it ends on (t within 1e-4 * max(1, t_max) of the ray's
    t_max: a shadow ray reaching its light) where the other does not see it. Returns (mismatch, near-tie, end-surface)
    masks; a mismatch in neither class is a traversal or intersector error."""
    t_gpu = np.where(h["prim"] == 0xFFFFFFFF, np.inf, h["t"])
    t_ref = np.where(ref_prim == 0xFFFFFFFF, np.inf, ref_t)
    mismatch = h["prim"] != ref_prim
    with np.errstate(invalid="ignore"):
        near_tie = mismatch & (np.abs(t_gpu - t_ref) <= 1e-4 * np.maximum(1.0, np.minimum(t_gpu, t_ref)))
        t_seen = np.minimum(t_gpu, t_ref)  # the hit only one side reports
        at_end = np.isfinite(ray_t_max) & (np.abs(t_seen - ray_t_max) <= 1e-4 * np.maximum(1.0, ray_t_max))
    ends = mismatch & ~near_tie & ((t_seen <= 1e-4) | at_end)
    return mismatch, near_tie, ends


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_closest_hit_ids_f32_near_tie_fraction(name):
    """fp32 device against the reference's answers (bvh_acceleration_structure.hpp:31-91) on (a) uniform random rays,
    (b) the rays an fp32 render casts (origins 1e-4 off the surfaces) and (c) the rays the reference itself logged,
    whose origins sit 1e-6 off a surface: below fp32 resolution at these coordinates, so in fp32 the origin may round
    onto or through its own surface.
    Bar (north_star: ids exact except a stated near-tie fraction): on (a) and (b) ids differ on <= 1e-3 of the rays
    and EVERY differing id is a near-tie in distance; on (c) every differing id is a near-tie or an end-surface case,
    near-ties <= 1e-3, end-surface cases <= 3e-2 (measured: 1.6e-2 on the room, 7e-4 on the zoo)."""
    z, scene = load_rays(name)
    nrnd = int(z["n_random"])
    with make_ctx(scene, "f32") as ctx:
        hits = ctx.trace_closest(z["rays"])
        anyh = ctx.trace_any(z["rays"])
        hits32 = ctx.trace_closest(z["rays32"])
        any32 = ctx.trace_any(z["rays32"])
        assert ctx.counters().stack_overflows == 0

    def check(h, ref_prim, ref_t, a, ref_a, label, start_bound, t_max, check_t=True):
        mismatch, near_tie, start = classify_id_mismatches(h, ref_prim, ref_t, t_max)
        unexplained = mismatch & ~near_tie & ~start
        print(f"{name} {label}: {len(ref_prim)} rays, id mismatches {mismatch.mean():.2e} = near-ties {near_tie.mean():.2e} + "
              f"end-surface {start.mean():.2e} + unexplained {int(unexplained.sum())}; any-hit mismatch {(a != ref_a).mean():.2e}")
        assert not unexplained.any(), [(int(i), int(h["prim"][i]), int(ref_prim[i]), float(h["t"][i]), float(ref_t[i])) for i in np.flatnonzero(unexplained)[:8]]
        assert near_tie.mean() <= 1e-3 and start.mean() <= start_bound
        assert (a != ref_a).mean() <= max(1e-3, start_bound)
        ok = (~mismatch) & (ref_prim != 0xFFFFFFFF)
        err = np.abs(h["t"][ok] - ref_t[ok]) / np.maximum(1.0, ref_t[ok])
        if check_t:  # (not for the reference's own rays: a start 1e-6 off a curved surface is not representable in fp32)
            assert np.quantile(err, 0.999) <= 1e-4 and err.max() <= 5e-2  # the tail is grazing incidence

    tm = z["rays"]["t_max"]
    check(hits[-nrnd:], z["bvh_prim"][-nrnd:], z["bvh_t"][-nrnd:], anyh[-nrnd:], z["any_hit"][-nrnd:], "random rays", 0.0, tm[-nrnd:])
    check(hits32, z["rays32_prim"], z["rays32_t"], any32, z["rays32_any"], "fp32-render rays", 0.0, z["rays32"]["t_max"])
    check(hits[:-nrnd], z["bvh_prim"][:-nrnd], z["bvh_t"][:-nrnd], anyh[:-nrnd], z["any_hit"][:-nrnd], "reference-logged rays", 3e-2, tm[:-nrnd], check_t=False)


@pytest.mark.parametrize("name", FILM_FIXTURES)
def test_radiance_f64_matches_reference(name):
    z, scene = load_film(name)
    with make_ctx(scene, "f64") as ctx:
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        for tag, integ in (("mis", A.INTEGRATOR_MIS), ("bsdf", A.INTEGRATOR_BSDF)):
            ctx.clear_film()
            ctx.reset_counters()
            ctx.render_pass(0, int(z["spp"]), integ, int(z["max_path_length"]), True)
            rgb, n = ctx.read_film()
            c = ctx.counters()
            ref = z[tag + "_rgb_sum"]
            assert n == int(z[tag + "_count"])
            rel = pixel_rel_err(rgb, ref)
            assert rel.max() <= 1e-9, (tag, rel.max())
            rays = z[tag + "_rays"]
            assert c.closest_rays == int(rays[0])
            # visibility rays are cast where bsdf->evaluate() is non-zero, a test that rounding decides where the light
            # direction lies in the surface (a vertex on a plane emitter sampling that same emitter: normals_scene has
            # 600 of them) or grazes it. The fp64 kernels are compiled without fused multiply-adds, like the reference
            # build, so nearly all of these decisions fall the same way (with contraction normals_mirror was 23 rays
            # off, now 0); what is left, at most 5 rays in 175 000 on these fixtures, is the device's libm against glibc
            assert abs(int(c.shadow_rays) - int(rays[1])) <= 8
            unoccluded, ref_unoccluded = int(c.shadow_rays) - int(c.shadow_occluded), int(rays[1]) - int(rays[2])
            assert abs(unoccluded - ref_unoccluded) <= 8
            assert c.nan_samples == int(rays[3]) and c.stack_overflows == 0


@pytest.mark.parametrize("name", FILM_FIXTURES)
def test_radiance_f32_within_stated_tolerance(name):
    z, scene = load_film(name)
    with make_ctx(scene, "f32") as ctx:
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        ctx.render_pass(0, int(z["spp"]), A.INTEGRATOR_MIS, int(z["max_path_length"]), True)
        rgb, n = ctx.read_film()
        assert ctx.counters().stack_overflows == 0
    ref = z["mis_rgb_sum"]
    rel = pixel_rel_err(rgb, ref)
    print(f"{name}: fp32 median rel err {np.median(rel):.2e}, within 1e-2: {(rel <= 1e-2).mean():.3f}")
    assert np.median(rel) <= 5e-6
    assert (rel <= 1e-2).mean() >= 0.99
    assert abs(rgb.mean() - ref.mean()) <= 0.005 * ref.mean()


@pytest.mark.parametrize("offset", ["along_ray", "normal"])
def test_f32_matches_f32_restatement_statistically(offset):
    """GPU fp32 vs the float instantiation of the CPU oracle (same epsilon, same streams), with the fp32 default origin
    offset (along the ray: the restatement mirrors that deviation, so this leg checks the device against a copy of its
    own rule) and with the reference's own rule (along the normal: the restatement then is the reference's algorithm in
    float). Measured: median 4.2e-7, 99.96 % within 1e-2, 98.8 % within 1e-4 in both. What bounds fp32 against the REAL
    reference are the film fixtures above (every scene class: zoo, glass, nested media, textures ...) and the crops of
    tests/test_gpu_headline.py."""
    scene = S.room(width=64, height=36, tess=4)
    o = pyref.restatement("f32").scene(scene, along_ray=offset == "along_ray")
    ref, n, _ = o.render(4, seed=2, threads=4)
    with make_ctx(scene, "f32", origin_offset=offset) as ctx:
        ctx.set_sampler(2)
        ctx.render_pass(0, 4)
        rgb, m = ctx.read_film()
    assert n == m
    rel = pixel_rel_err(rgb, ref)
    print(f"fp32 device vs fp32 restatement ({offset}): median {np.median(rel):.2e}, within 1e-2 {(rel <= 1e-2).mean():.4f}, within 1e-4 {(rel <= 1e-4).mean():.4f}")
    assert np.median(rel) <= 2e-6 and (rel <= 1e-2).mean() >= 0.998 and (rel <= 1e-4).mean() >= 0.97


def test_converged_image_relmse():
    """relMSE(GPU fp32, fp64 oracle) at higher sample counts stays at the Monte-Carlo noise floor."""
    scene = S.room(width=48, height=27, tess=4)
    o = pyref.restatement("f64").scene(scene)
    ref, n, _ = o.render(256, seed=11, threads=8)
    ref = ref / n
    with make_ctx(scene, "f32") as ctx:
        ctx.set_sampler(11)
        ctx.render_pass(0, 256)
        rgb, m = ctx.read_film()
    img = rgb / m
    relmse = np.mean((img - ref) ** 2 / (ref ** 2 + 1e-4))
    print("relMSE, same streams, 256 spp:", relmse)
    assert relmse <= 2e-3


# ------------------------------------------------------------------ the other hierarchy builders (FCGPU_BVH_DEVICE_LBVH,
# FCGPU_BVH_HOST_REFERENCE; every other test runs on the default, FCGPU_BVH_HOST_SAH)
@pytest.mark.parametrize("builder", ["device", "reference"])
@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_other_builders_return_the_reference_hits(name, builder):
    """Any valid BVH finds the same closest hit: the linear BVH built on the GPU and the host build with the
    reference's own split rule must give the reference's primitive ids and distances on the golden ray sets."""
    z, scene = load_rays(name)
    with make_ctx(scene, "f64", bvh_builder=builder) as ctx:
        hits = ctx.trace_closest(z["rays"])
        anyh = ctx.trace_any(z["rays"])
        c = ctx.counters()
    assert c.stack_overflows == 0
    if builder == "device":
        assert c.reserved[2] == 2 * scene.primitive_total() - 1  # one primitive per binary leaf
    ref_prim = z["bvh_prim"]
    mismatch = hits["prim"] != ref_prim
    assert mismatch.mean() <= 1e-4, f"prim id mismatches {mismatch.sum()} / {len(ref_prim)}"
    ok = (~mismatch) & (ref_prim != 0xFFFFFFFF)
    assert np.allclose(hits["t"][ok], z["bvh_t"][ok], rtol=1e-9, atol=1e-9)
    assert (anyh != z["any_hit"]).mean() <= 1e-4


@pytest.mark.parametrize("builder", ["device", "reference"])
@pytest.mark.parametrize("name", ["room_t4", "zoo", "mask", "nested_rough"])
def test_other_builders_radiance_f64_matches_reference(name, builder):
    z, scene = load_film(name)
    with make_ctx(scene, "f64", bvh_builder=builder) as ctx:
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        ctx.render_pass(0, int(z["spp"]), A.INTEGRATOR_MIS, int(z["max_path_length"]), True)
        rgb, n = ctx.read_film()
        c = ctx.counters()
    assert n == int(z["mis_count"]) and c.stack_overflows == 0
    assert pixel_rel_err(rgb, z["mis_rgb_sum"]).max() <= 1e-9
    assert c.closest_rays == int(z["mis_rays"][0])


# ------------------------------------------------------------------ the 8-wide hierarchy (fcgpu_options.bvh_width = 8)
@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_eight_wide_hierarchy_returns_the_reference_hits(name):
    """Node8 (80-byte node, octant-ordered child slots, sphere / plane leaves as proxy records): same closest hits, distances
    and occlusion answers as the reference on the golden ray sets, in both precisions' WIDE8 kernels."""
    z, scene = load_rays(name)
    with make_ctx(scene, "f64", bvh_width=8) as ctx:
        hits = ctx.trace_closest(z["rays"])
        anyh = ctx.trace_any(z["rays"])
        assert ctx.counters().stack_overflows == 0
    mismatch = hits["prim"] != z["bvh_prim"]
    assert mismatch.mean() <= 1e-4, f"prim id mismatches {mismatch.sum()} / {len(mismatch)}"
    ok = (~mismatch) & (z["bvh_prim"] != 0xFFFFFFFF)
    assert np.allclose(hits["t"][ok], z["bvh_t"][ok], rtol=1e-9, atol=1e-9)
    assert (anyh != z["any_hit"]).mean() <= 1e-4
    with make_ctx(scene, "f32", bvh_width=8) as ctx:
        h32 = ctx.trace_closest(z["rays32"])
        assert ctx.counters().stack_overflows == 0
    mismatch, near_tie, ends = classify_id_mismatches(h32, z["rays32_prim"], z["rays32_t"], z["rays32"]["t_max"])
    assert not (mismatch & ~near_tie & ~ends).any() and near_tie.mean() <= 1e-3


@pytest.mark.parametrize("name", ["room_t4", "zoo", "mask", "nested_rough", "glass", "textured", "nested_deep9"])
def test_eight_wide_hierarchy_radiance_f64_matches_reference(name):
    z, scene = load_film(name)
    with make_ctx(scene, "f64", bvh_width=8) as ctx:
        ctx.set_sampler(int(z["seed"]), stratified=film_stratified(z))
        ctx.render_pass(0, int(z["spp"]), A.INTEGRATOR_MIS, int(z["max_path_length"]), True)
        rgb, n = ctx.read_film()
        c = ctx.counters()
    assert n == int(z["mis_count"]) and c.stack_overflows == 0 and c.medium_overflows == 0
    assert pixel_rel_err(rgb, z["mis_rgb_sum"]).max() <= 1e-9
    assert c.closest_rays == int(z["mis_rays"][0])


def test_eight_wide_hierarchy_refuses_what_its_stack_cannot_hold():
    from common import ribbon_scene

    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32", bvh_width=8)
        with pytest.raises(R.FcgpuError) as e:
            ctx.upload_scene(ribbon_scene(420))
        assert e.value.code == A.E_LIMIT and "bvh_width 4" in str(e.value)
        with pytest.raises(R.FcgpuError) as e:
            ctx.set_options(precision="f32", bvh_width=6)
        assert e.value.code == A.E_INVALID


def test_all_builders_agree_on_a_large_mesh():
    """196 608 wall triangles in fp32: the three hierarchies cast the same rays and render the same image up to the few
    paths whose exact-tie hits (shared triangle edges) resolve to the neighbouring triangle; the default build takes
    the fewest node steps per ray."""
    scene = S.room(width=160, height=90, tess=128)
    out = {}
    for builder in ("host", "device", "reference"):
        with make_ctx(scene, "f32", bvh_builder=builder) as ctx:
            ctx.set_sampler(5)
            ctx.render_pass(0, 4)
            rgb, n = ctx.read_film()
            c = ctx.counters()
            out[builder] = (rgb / n, c.closest_rays, c.reserved[3] / (c.closest_rays + c.shadow_rays), c.reserved[6])
            assert c.stack_overflows == 0 and c.nan_samples == 0
    print({k: dict(node_steps_per_ray=round(v[2], 2), flatten_ms=v[3] / 1e3) for k, v in out.items()})
    for other in ("device", "reference"):
        rel = pixel_rel_err(out[other][0], out["host"][0])
        assert np.median(rel) <= 1e-6 and (rel <= 1e-2).mean() >= 0.98, other
        assert abs(int(out["host"][1]) - int(out[other][1])) <= 1e-3 * out["host"][1], other
        assert out["host"][2] < out[other][2], other


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_f64_match_the_restatement(seed):
    """Differential test: seeded random assemblies of every supported feature (scenes.random_scene; the restatement
    equals the real reference on them, tests/test_oracle_vs_reference.py), GPU fp64 against the restatement."""
    scene = S.random_scene(seed, 32, 24)
    o = pyref.restatement("f64").scene(scene)
    with make_ctx(scene, "f64") as ctx:
        ctx.set_sampler(seed)
        for integ in (A.INTEGRATOR_MIS, A.INTEGRATOR_BSDF):
            ref, n, st = o.render(3, integrator=integ, max_path_length=8, seed=seed, threads=4)
            ctx.clear_film()
            ctx.reset_counters()
            ctx.render_pass(0, 3, integ, 8, True)
            rgb, m = ctx.read_film()
            c = ctx.counters()
            assert m == n and c.stack_overflows == 0
            assert pixel_rel_err(rgb, ref).max() <= 1e-9, (seed, integ)
            assert c.closest_rays == st.closest_rays and c.nan_samples == st.nan_samples
            assert abs(int(c.shadow_rays) - int(st.any_rays)) <= 8


@pytest.mark.parametrize("seed", range(12, 18))
def test_random_scenes_stratified_on_a_device_built_bvh(seed):
    """Same differential test with the stratified stream and the hierarchy built on the GPU."""
    scene = S.random_scene(seed, 28, 20)
    o = pyref.restatement("f64").scene(scene)
    ref, n, st = o.render(4, max_path_length=6, seed=seed, threads=4, stratified=4)
    with make_ctx(scene, "f64", bvh_builder="device") as ctx:
        ctx.set_sampler(seed, stratified=4)
        ctx.render_pass(0, 4, A.INTEGRATOR_MIS, 6, True)
        rgb, m = ctx.read_film()
        c = ctx.counters()
    assert m == n and c.stack_overflows == 0
    assert pixel_rel_err(rgb, ref).max() <= 1e-9
    assert c.closest_rays == st.closest_rays and c.nan_samples == st.nan_samples
    assert abs(int(c.shadow_rays) - int(st.any_rays)) <= 8


# ------------------------------------------------------------------ per-lobe tables on the device (fcgpu_bsdf_table)
def _bsdf_table_cases():
    z = np.load(os.path.join(GOLDEN, "bsdf_tables.npz"))
    names = [str(n) for n in z["material_names"]]
    stored = np.frombuffer(z["materials"].tobytes(), dtype=np.uint8).reshape(len(names), -1)
    mats = np.zeros((len(names), C.sizeof(A.Material)), np.uint8)
    mats[:, : stored.shape[1]] = stored  # the fixture predates the tex_* handles: zero = constant parameter
    for mi, name in enumerate(names):
        m = A.Material.from_buffer_copy(mats[mi].tobytes())
        for ei, (ea, eb) in enumerate(z["etas"]):
            for fi, fr in enumerate(z["frames"]):
                yield name, ei, fi, m, float(ea), float(eb), fr, z[f"t|{name}|{ei}|{fi}"], z["queries"]


def _singular(ref_rows):
    """Rows where the reference itself returns rounding noise amplified without bound: the refraction Jacobian
    -o.h / (eta i.h + o.h)^2 (microfacet_glass.hpp:54, microfacet_transmission.hpp:39) with eta_a == eta_b, whose
    denominator is exactly zero in exact arithmetic (values of 1e31..inf). Nothing can be compared there."""
    with np.errstate(invalid="ignore"):
        return ~np.isfinite(ref_rows).all(axis=1) | (np.abs(ref_rows).max(axis=1) > 1e20)


def test_device_bsdf_tables_f64_match_reference():
    """bsdf::evaluate / pdf_wi / sample_wi of every material class computed by the device functions the shade kernels
    call, in fp64, against the REAL reference's tables (tests/golden/bsdf_tables.npz: 512 queries incl. grazing
    directions at |cos| = 1e-3, four eta pairs incl. (1.3614, 1.52) and (1, 1), an orthonormal and a sphere-like skewed
    frame). Bar: discrete outputs (picked lobe, delta flag, sample success) identical; every real field within 1e-10
    relative, or 1e-12 of the field's typical magnitude where the exact value is zero (Fresnel between equal indices:
    the reference returns 1e-31, i.e. its own rounding noise); singular rows (see _singular) must be singular here too."""
    checked = singular = 0
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f64")
        for name, ei, fi, m, ea, eb, fr, ref, q in _bsdf_table_cases():
            got = ctx.bsdf_table(m, q, ng=fr[0], tangent=fr[1], ns=fr[2] / np.linalg.norm(fr[2]), bitangent=fr[3], eta_a=ea, eta_b=eb)
            for field in ("lobe", "is_delta", "s_ok"):
                assert np.array_equal(got[field], ref[field]), (name, ei, fi, field)
            for field in ("f", "pdf", "s_wi", "s_value", "s_pdf"):
                a, b = got[field].reshape(len(q), -1), ref[field].reshape(len(q), -1)
                sing = _singular(b)
                singular += int(sing.sum())
                with np.errstate(invalid="ignore"):
                    assert (~np.isfinite(a[sing]).all(axis=1) | (np.abs(a[sing]).max(axis=1) > 1e20)).all(), (name, ei, fi, field)
                scale = np.median(np.abs(b[~sing][np.abs(b[~sing]) > 1e-20])) if (np.abs(b[~sing]) > 1e-20).any() else 0.0
                assert np.allclose(a[~sing], b[~sing], rtol=1e-10, atol=1e-12 * scale), (name, ei, fi, field)
            checked += len(q)
    assert checked == 12 * 4 * 2 * 512
    print(f"singular (index-matched refraction) field rows skipped: {singular}")


def test_device_bsdf_tables_f32_within_stated_bounds():
    """The same tables from the fp32 product functions. Error of a field on a row = largest component difference /
    max(largest component of the reference, 1e-3 x the field's median magnitude in that table), i.e. a vector error for
    f, s_value and the unit vector s_wi; singular rows (see _singular) skipped. Stated bounds over all tables:
      * picked lobe and delta flag: identical (the pick cdf stays double on the device);
      * sample success flag: differs on at most 1e-3 of the queries (a sampled direction at the horizon, total internal
        reflection decided by the last bit);
      * ordinary rows: median <= 5e-7, 99.9th percentile <= 2e-4 (measured: <= 1e-7, <= 2e-5);
      * hard rows -- a grazing direction (|cos| = 1e-3, where the smooth GGX lobes' alpha = 4e-6 puts D near 1e10) or
        the index pairs (1.3614, 1.52) and (1, 1), whose refraction Jacobian cancels --: median <= 5e-7, 99th percentile
        <= 1e-4 (measured 6e-6), worst row <= 0.2 (measured 8e-2: rough glass at grazing incidence)."""
    errs = {f: [] for f in ("f", "pdf", "s_wi", "s_value", "s_pdf")}
    hard = {f: [] for f in errs}
    worst = {f: (0.0, None) for f in errs}
    flips = total = 0
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f32")
        for name, ei, fi, m, ea, eb, fr, ref, q in _bsdf_table_cases():
            got = ctx.bsdf_table(m, q, ng=fr[0], tangent=fr[1], ns=fr[2] / np.linalg.norm(fr[2]), bitangent=fr[3], eta_a=ea, eta_b=eb)
            assert np.array_equal(got["lobe"], ref["lobe"]) and np.array_equal(got["is_delta"], ref["is_delta"]), (name, ei, fi)
            flips += int((got["s_ok"] != ref["s_ok"]).sum())
            total += len(q)
            grazing = (np.abs(q["wo"][:, 1]) < 2e-3) | (np.abs(q["wi"][:, 1]) < 2e-3)
            both = (got["s_ok"] == 1) & (ref["s_ok"] == 1)
            for field in errs:
                a, b = got[field].reshape(len(q), -1), ref[field].reshape(len(q), -1)
                rows = (both if field.startswith("s_") else np.ones(len(q), bool)) & ~_singular(b)
                mag = np.abs(b[rows]).max(axis=1) if rows.any() else np.zeros(0)
                floor = 1e-3 * (np.median(mag[mag > 1e-20]) if (mag > 1e-20).any() else 1.0)
                with np.errstate(invalid="ignore"):
                    rel = np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), floor)
                rel = np.where(np.isfinite(rel), rel, np.inf)
                is_hard = grazing | (ei >= 2)
                errs[field].append(rel[rows & ~is_hard])
                hard[field].append(rel[rows & is_hard])
                if rows.any() and rel[rows].max() > worst[field][0]:
                    k = int(np.flatnonzero(rows)[np.argmax(rel[rows])])
                    worst[field] = (float(rel[k]), (name, ei, fi, k, a[k].tolist(), b[k].tolist()))
    print(f"sample-success flips: {flips} of {total}")
    for field in errs:
        e, h = np.concatenate(errs[field]), np.concatenate(hard[field])
        print(f"{field:8s} ordinary rows ({len(e)}): median {np.median(e):.2e} p99.9 {np.quantile(e, 0.999):.2e} max {e.max():.2e} | "
              f"hard rows ({len(h)}): median {np.median(h):.2e} p99 {np.quantile(h, 0.99):.2e} max {h.max():.2e} | worst {worst[field][1]}")
    assert flips <= 1e-3 * total
    for field in errs:
        e, h = np.concatenate(errs[field]), np.concatenate(hard[field])
        assert np.median(e) <= 5e-7 and np.quantile(e, 0.999) <= 2e-4, field
        assert np.median(h) <= 5e-7 and np.quantile(h, 0.99) <= 1e-4 and h.max() <= 0.2, field
