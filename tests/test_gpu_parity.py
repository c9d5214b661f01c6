"""GPU parity tests proper: the HIP path (through the C-ABI) against the golden vectors produced
by the real reference and against the CPU oracle on identical sample streams.

Bars (stated here, repeated in DESIGN.md):
  * fp64 mode: per-pixel radiance within 1e-9 relative of the reference (same algorithm, same
    precision; only libm ulps and FMA contraction differ); closest-hit primitive ids identical on
    the dumped ray sets; ray counts equal up to 1e-3 relative.
  * fp32 mode (the fast path): closest-hit ids identical except a near-tie fraction <= 1e-3, with
    |t_gpu - t_ref| <= 1e-4 * max(1, t_ref); per-pixel radiance on identical streams:
    median relative error <= 1e-4 and >= 90 % of pixels within 1e-2 at 4 spp (individual paths
    legitimately diverge after a discrete decision flips; those pixels are not bounded), image
    mean within 2 %.
"""
import numpy as np
import pytest

from common import A, FILM_FIXTURES, R, RAY_FIXTURES, S, film_stratified, load_film, load_rays, pixel_rel_err
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


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_closest_hit_ids_f32_near_tie_fraction(name):
    """fp32 device against the reference's answers on (a) uniform random rays and (b) the rays an fp32
    render casts (origins 1e-4 off the surfaces). The reference's own logged rays start 1e-6 off a
    surface, which fp32 cannot represent at these coordinates: every one is a near-tie, so for them
    only a loose bound is asserted."""
    z, scene = load_rays(name)
    nrnd = int(z["n_random"])
    with make_ctx(scene, "f32") as ctx:
        hits = ctx.trace_closest(z["rays"])
        anyh = ctx.trace_any(z["rays"])
        hits32 = ctx.trace_closest(z["rays32"])
        any32 = ctx.trace_any(z["rays32"])
        assert ctx.counters().stack_overflows == 0

    def check(h, ref_prim, ref_t, a, ref_a, bound, label, check_t=True):
        mismatch = h["prim"] != ref_prim
        print(f"{name} {label}: prim-id mismatch {mismatch.mean():.2e}, any-hit mismatch {(a != ref_a).mean():.2e}")
        assert mismatch.mean() <= bound and (a != ref_a).mean() <= bound
        ok = (~mismatch) & (ref_prim != 0xFFFFFFFF)
        err = np.abs(h["t"][ok] - ref_t[ok]) / np.maximum(1.0, ref_t[ok])
        if check_t:
            assert np.quantile(err, 0.999) <= 1e-4 and err.max() <= 5e-2  # the tail is grazing incidence

    check(hits[-nrnd:], z["bvh_prim"][-nrnd:], z["bvh_t"][-nrnd:], anyh[-nrnd:], z["any_hit"][-nrnd:], 1e-3, "random rays")
    check(hits32, z["rays32_prim"], z["rays32_t"], any32, z["rays32_any"], 1e-3, "fp32-render rays")
    check(hits[:-nrnd], z["bvh_prim"][:-nrnd], z["bvh_t"][:-nrnd], anyh[:-nrnd], z["any_hit"][:-nrnd], 3e-2, "reference-logged rays (all near-ties in fp32)", check_t=False)


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
    assert np.median(rel) <= 1e-4
    assert (rel <= 1e-2).mean() >= 0.90
    assert abs(rgb.mean() - ref.mean()) <= 0.02 * ref.mean()


def test_f32_matches_f32_restatement_statistically():
    """GPU fp32 vs the float instantiation of the CPU oracle (same epsilon, same streams)."""
    scene = S.room(width=64, height=36, tess=4)
    o = pyref.restatement("f32").scene(scene)
    ref, n, _ = o.render(4, seed=2, threads=4)
    with make_ctx(scene, "f32") as ctx:
        ctx.set_sampler(2)
        ctx.render_pass(0, 4)
        rgb, m = ctx.read_film()
    assert n == m
    rel = pixel_rel_err(rgb, ref)
    assert np.median(rel) <= 1e-4 and (rel <= 1e-2).mean() >= 0.9


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
