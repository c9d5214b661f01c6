"""The CPU restatement (oracle/liboracle.so) against the golden vectors that the REAL reference
produced (oracle/make_golden.py). This is what pins the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from common import A, FILM_FIXTURES, GOLDEN, RAY_FIXTURES, film_stratified, load_film, load_rays
from oracle import pyref


@pytest.fixture(scope="module")
def oracle():
    return pyref.restatement("f64")


@pytest.mark.parametrize("name", FILM_FIXTURES)
def test_film_matches_reference(oracle, name):
    z, scene = load_film(name)
    sc = oracle.scene(scene)
    for tag, integ in (("mis", A.INTEGRATOR_MIS), ("bsdf", A.INTEGRATOR_BSDF)):
        rgb, n, st = sc.render(int(z["spp"]), integrator=integ, max_path_length=int(z["max_path_length"]), seed=int(z["seed"]), threads=4,
                               stratified=film_stratified(z))
        ref = z[tag + "_rgb_sum"]
        assert n == int(z[tag + "_count"])
        # same algorithm in the same precision: only the order of a few double roundings differs
        assert np.abs(rgb - ref).max() <= 1e-9 * np.abs(ref).max()
        rays = z[tag + "_rays"]
        assert (st.closest_rays, st.any_rays, st.any_occluded, st.nan_samples) == tuple(int(v) for v in rays)


@pytest.mark.parametrize("name", RAY_FIXTURES)
def test_hits_match_reference(oracle, name):
    z, scene = load_rays(name)
    sc = oracle.scene(scene)
    rays = z["rays"]
    hits = sc.trace_closest(rays)
    assert np.array_equal(hits["prim"], z["bvh_prim"])  # bit-exact ids, ties included (same traversal order)
    hit = z["bvh_prim"] != 0xFFFFFFFF
    assert np.allclose(hits["t"][hit], z["bvh_t"][hit], rtol=1e-12, atol=1e-12)
    brute = sc.trace_closest(rays, brute_force=True)
    assert np.array_equal(brute["prim"], z["brute_prim"])
    assert np.array_equal(sc.trace_any(rays), z["any_hit"])
    assert np.array_equal(sc.trace_any(rays, brute_force=True), z["any_hit"])


def test_bsdf_tables_match_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "bsdf_tables.npz"))
    names = [str(n) for n in z["material_names"]]
    # the fixture holds fcgpu_material records as they were when it was written (the tex_* handles were appended
    # later: zero = constant parameter, which is what the tables are about)
    stored = np.frombuffer(z["materials"].tobytes(), dtype=np.uint8).reshape(len(names), -1)
    mats = np.zeros((len(names), C.sizeof(A.Material)), np.uint8)
    mats[:, : stored.shape[1]] = stored
    q = z["queries"]
    checked = 0
    for mi, name in enumerate(names):
        m = A.Material.from_buffer_copy(mats[mi].tobytes())
        for ei, (ea, eb) in enumerate(z["etas"]):
            for fi, fr in enumerate(z["frames"]):
                ref = z[f"t|{name}|{ei}|{fi}"]
                ns = fr[2] / np.linalg.norm(fr[2])
                got = oracle.bsdf_table(m, q, ng=fr[0], tangent=fr[1], ns=ns, bitangent=fr[3], eta_a=ea, eta_b=eb)
                assert np.array_equal(got["lobe"], ref["lobe"]) and np.array_equal(got["is_delta"], ref["is_delta"])
                assert np.array_equal(got["s_ok"], ref["s_ok"])
                for field in ("f", "pdf", "s_wi", "s_value", "s_pdf"):
                    a, b = got[field], ref[field]
                    assert np.allclose(a, b, rtol=1e-10, atol=1e-300, equal_nan=True), (name, ei, fi, field)
                checked += len(q)
    assert checked == len(names) * 4 * 2 * len(q)


def test_sampler_known_answers():
    """fcgpu_sampler.h (compiled into the oracle) against integers computed in pure Python."""
    kat = np.load(os.path.join(GOLDEN, "sampler_kat.npy"))
    M = (1 << 64) - 1

    def mix64(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    for seed, px, py, s, dim, x24, y24 in kat.tolist():
        key = mix64(mix64(px | (py << 20) | (s << 40)) ^ ((seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & M))
        bits = mix64((key + (dim + 1) * 0x9E3779B97F4A7C15) & M)
        assert (bits >> 40, (bits & 0xFFFFFFFF) >> 8) == (x24, y24)


def test_f32_restatement_is_close_to_f64(oracle):
    """The float instantiation defines what an fp32 device should produce: statistically the same image."""
    z, scene = load_film("room_t4")
    a, n, _ = pyref.restatement("f32").scene(scene).render(int(z["spp"]), seed=int(z["seed"]), threads=4)
    ref = z["mis_rgb_sum"]
    assert abs(a.mean() - ref.mean()) < 0.02 * ref.mean()


def test_ulp_noise_probe_is_off_by_default_and_small_when_on(oracle):
    """tools/ulp_sensitivity.py's hook: transcendental results moved by -1 / 0 / +1 ulp. Off: the fixture bit for bit as
    before; on: the image moves by rounding noise only (a pixel that moves by more is ill-conditioned, which is what the tool
    is for), and switching it off restores the plain result."""
    z, scene = load_film("zoo")
    sc = oracle.scene(scene)
    args = dict(max_path_length=int(z["max_path_length"]), seed=int(z["seed"]), threads=4, stratified=film_stratified(z))
    plain, _, _ = sc.render(int(z["spp"]), **args)
    oracle.set_ulp_noise(7)
    try:
        noisy, _, _ = sc.render(int(z["spp"]), **args)
    finally:
        oracle.set_ulp_noise(0)
    again, _, _ = sc.render(int(z["spp"]), **args)
    assert np.array_equal(plain, again)
    assert not np.array_equal(plain, noisy)
    assert np.abs(noisy - plain).max() <= 1e-9 * np.abs(plain).max()
