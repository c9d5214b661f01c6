"""Parity AT the BASELINE configurations' own sizes (BASELINE.json configs[1..3]), not on thumbnails: crops of the full
frames against the REAL reference on identical sample streams (tests/golden/crop_*.npz, written by oracle/make_golden.py:
CROPS). Round 1's fp32 origin-offset defect was invisible on the 40 x 24 fixtures and only showed at tess 64.

  headline           room, Forward MIS, 1920x1080, tess 64, centre sphere rough glass; centre 240 x 135, 16 spp
  c3_smooth_mirror,  the same frame with the other centre materials of the sweep: delta mirror, delta glass + its medium,
  c3_smooth_glass,   the two-lobe plastic
  c3_plastic
  c4_mask            mask_scene(detail 256 = 255 520 triangles, four 1024^2 maps, thin lens, texture environment) at
                     3840x2160: the SPILL + TEX + normal-mapping instantiations; the crop crosses the silhouette and an eye
                     opening, 8 spp

Bars:
  fp64: every pixel within 1e-9, closest-ray count equal, shadow-ray count within 1e-4, NaN count equal;
  fp32 (the product path), per crop in F32_BARS below = measured value + margin (a pixel is the sum of its paths, and one
        path whose discrete decision flips on a rounding moves it): median per-pixel relative error, fraction of pixels
        within 1e-2 and within 1e-3, image mean, ray counts;
  and the fp32 residue must be what DESIGN.md section 4 says it is: the decision logs of an fp32 and an fp64 render of the
  crop are compared path by path (common.divergence_table): bounded fraction of diverged paths, nearly all of them a
  closest hit on another primitive, and every pixel beyond 1e-2 holds a diverged path or a visibility difference."""
import os

import numpy as np
import pytest

import bench
from common import CROP_FIXTURES, divergence_table, load_crop, pixel_rel_err, render_crop

pytestmark = pytest.mark.gpu

# crop -> (median rel err <=, within 1e-2 >=, within 1e-3 >=, |mean ratio - 1| <=); measured on MI355X (round 3) in the comments.
# The headline crop's bar IS bench.py's parity gate (one bar, stated there). Round 3 solved the hit point of an accepted SPHERE
# hit in double (fc_traverse.h: sphere_point_f64): the fp32 quadratic left 4e-6 in t, which the curved spheres of the room
# turned into hits on the wrong side of a triangle edge one bounce later; before that the same crops measured 0.989 / 0.957 /
# 0.985 / 0.984 within 1e-2. The smooth mirror stays the hardest: every path of the crop is mirrored by the sphere onto the walls.
F32_BARS = {
    "headline": (bench.PARITY_MEDIAN_REL_ERR, bench.PARITY_WITHIN_1E2_MIN_16SPP, bench.PARITY_WITHIN_1E3_MIN_16SPP, bench.PARITY_MEAN_RATIO),
                                                     # 3.9e-7, 0.9970, 0.9923, 1.6e-5
    "c3_smooth_mirror": (2e-6, 0.99, 0.978, 3e-4),   # 5.1e-7, 0.9940, 0.9846, 1.0e-5
    "c3_smooth_glass": (2e-6, 0.994, 0.987, 3e-4),   # 3.1e-7, 0.9969, 0.9919, 4e-6
    "c3_plastic": (2e-6, 0.994, 0.986, 3e-4),        # 4.1e-7, 0.9970, 0.9906, 1.7e-5
    "c4_mask": (4e-6, 0.988, 0.945, 5e-4),           # 1.1e-6, 0.9924, 0.9549, 1.7e-4
}
# crop -> (diverged paths <=, share of first divergences that are a closest hit on another primitive >=, pixels beyond 1e-2
# that hold neither a diverged path nor a visibility difference <=); measured in the comments.
# c4_mask: its 160 unattributed pixels are not decisions but the roughness MAP: an fp32 camera ray lands ~1e-4 texels from the
# fp64 one, the 1024^2 roughness map has 35 % texel-to-texel contrast (noise), and GGX's D follows roughness to the
# fourth power: ablation per texture in DESIGN.md section 4 (only the roughness and normal maps reproduce it, with a box
# filter it is gone; double-precision texture coordinates and barycentrics do not remove it: the ray itself is the fp32 quantity).
DIVERGENCE_BARS = {
    "headline": (3e-3, 0.95, 0),            # 1.94e-3, 0.973, 0
    "c3_smooth_mirror": (6e-3, 0.97, 0),    # 4.27e-3, 0.989, 0
    "c4_mask": (2e-3, 0.85, 260),           # 7.3e-4, 0.937, 160 of 32 400
}


def test_every_crop_fixture_has_bars():
    assert sorted(F32_BARS) == CROP_FIXTURES


@pytest.mark.parametrize("name", CROP_FIXTURES)
def test_crop_f64_matches_reference(name):
    z, scene, tile = load_crop(name)
    rgb, n, c = render_crop(scene, z, tile, "f64")
    assert n == int(z["mis_count"]) and c.stack_overflows == 0 and c.medium_overflows == 0
    assert pixel_rel_err(rgb, z["mis_rgb_sum"]).max() <= 1e-9
    # visibility rays are cast where bsdf->evaluate() is non-zero, a test rounding decides at grazing light directions (device
    # libm against glibc): 160 of 4.4 M rays differ on the headline crop, all of them occluded or carrying zero radiance
    assert c.closest_rays == int(z["mis_rays"][0]) and abs(int(c.shadow_rays) - int(z["mis_rays"][1])) <= max(8, 1e-4 * int(z["mis_rays"][1]))
    assert c.nan_samples == int(z["mis_rays"][3])


@pytest.mark.parametrize("name", CROP_FIXTURES)
def test_crop_f32_within_stated_tolerance(name):
    z, scene, tile = load_crop(name)
    rgb, n, c = render_crop(scene, z, tile, "f32")
    ref = z["mis_rgb_sum"]
    rel = pixel_rel_err(rgb, ref)
    print(f"crop {name} fp32: median rel err {np.median(rel):.2e}, within 1e-2: {(rel <= 1e-2).mean():.4f}, within 1e-3: {(rel <= 1e-3).mean():.4f}, "
          f"mean ratio {rgb.mean() / ref.mean():.6f}, closest rays {int(c.closest_rays)} vs {int(z['mis_rays'][0])}")
    assert n == int(z["mis_count"]) and c.stack_overflows == 0 and c.medium_overflows == 0 and c.nan_samples == int(z["mis_rays"][3])
    median, within2, within3, mean_tol = F32_BARS[name]
    assert np.median(rel) <= median
    assert (rel <= 1e-2).mean() >= within2
    assert (rel <= 1e-3).mean() >= within3
    assert abs(rgb.mean() / ref.mean() - 1.0) <= mean_tol
    # the rays cast: equal up to the paths that diverged
    assert abs(int(c.closest_rays) - int(z["mis_rays"][0])) <= 1e-3 * int(z["mis_rays"][0])


@pytest.mark.parametrize("name", sorted(DIVERGENCE_BARS))
def test_crop_f32_residue_is_attributed(name):
    """What separates fp32 from fp64 on the same samples is discrete decisions flipped by rounding, not a bias: gated."""
    z, scene, tile = load_crop(name)
    img64, _, _, h64, f64 = render_crop(scene, z, tile, "f64", decision_log=True)
    img32, _, _, h32, f32 = render_crop(scene, z, tile, "f32", decision_log=True)
    lines = []
    d = divergence_table(h64, f64, img64, h32, f32, img32, tile[2] * tile[3], out=lines.append)
    print("\n".join(lines))
    max_diverged, min_other_prim, max_unexplained = DIVERGENCE_BARS[name]
    assert d["diverged_fraction"] <= max_diverged
    assert d["first_is_another_primitive"] >= min_other_prim
    assert d["bad_pixels_unexplained"] <= max_unexplained
    assert d["clean_median_rel_err"] <= 2e-6
    assert pixel_rel_err(img64, z["mis_rgb_sum"]).max() <= 1e-9  # the fp64 side of the comparison IS the reference
