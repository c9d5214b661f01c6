"""Restatement vs the real reference run live (only where oracle/_ref/libfcref.so exists)."""
import numpy as np
import pytest

from common import A, S
from oracle import pyref

pytestmark = pytest.mark.skipif(not pyref.have_reference(), reason="oracle/_ref/libfcref.so not built (reference tree absent)")

CASES = [
    ("room", dict(width=40, height=24, tess=3), 3),
    ("room", dict(width=32, height=18, tess=2, env=(0.3, 0.4, 0.5)), 3),
    ("primitives_zoo", dict(width=40, height=40), 3),
    ("nested_media", dict(width=32, height=32, smooth=False), 3),
]


@pytest.mark.parametrize("builder,kw,spp", CASES)
def test_render_identical(builder, kw, spp):
    scene = S.BUILDERS[builder](**kw)
    ref = pyref.reference().scene(scene)
    orc = pyref.restatement("f64").scene(scene)
    for integ in (A.INTEGRATOR_MIS, A.INTEGRATOR_BSDF):
        for mpl, vis in ((10, True), (4, False)):
            a, an, ast = ref.render(spp, integrator=integ, max_path_length=mpl, visible_env=vis, seed=5, threads=4)
            b, bn, bst = orc.render(spp, integrator=integ, max_path_length=mpl, visible_env=vis, seed=5, threads=4)
            assert an == bn
            assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
            assert (ast.closest_rays, ast.any_rays, ast.any_occluded) == (bst.closest_rays, bst.any_rays, bst.any_occluded)


def test_tile_and_sample_offsets():
    scene = S.room(width=32, height=18, tess=2)
    ref = pyref.reference().scene(scene)
    orc = pyref.restatement("f64").scene(scene)
    a, _, _ = ref.render(2, sample_begin=3, tile=(8, 4, 16, 10), seed=9)
    b, _, _ = orc.render(2, sample_begin=3, tile=(8, 4, 16, 10), seed=9)
    assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
    assert a[:4].max() == 0 and a[4:14, 8:24].min() > 0


@pytest.mark.parametrize("builder,kw", [("room", dict(width=32, height=18, tess=2)), ("mask_scene", dict(width=24, height=36, detail=12, texture_size=16)),
                                        ("normal_mapped_scene", dict(width=36, height=24))])
def test_drop_in_integrator_behind_the_reference_interface(builder, kw):
    """`gpu_forward_mis_integrator : fc::integrator` (oracle/ref_harness.cpp) driven by the reference's own
    pixel loop, perspective_camera measurement, render_target film and sampler object; its run_once runs the
    restatement's bounce loop. Same images and ray counts as the stock integrators => the interface is honoured
    (sample order through fc::sampler::get / advance_dimension included), incl. the stratified stream."""
    scene = S.BUILDERS[builder](**kw)
    ref = pyref.reference().scene(scene)
    orc = pyref.restatement("f64").scene(scene)
    for integ in (A.INTEGRATOR_MIS, A.INTEGRATOR_BSDF):
        for strat in (None, 4):
            a, an, ast = ref.render(4, integrator=integ, max_path_length=8, seed=11, threads=2, stratified=strat)
            b, bn, bst = ref.render(4, integrator=integ, max_path_length=8, seed=11, threads=2, stratified=strat, drop_in=orc)
            assert an == bn and a.max() > 0
            assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
            assert (ast.closest_rays, ast.any_rays, ast.any_occluded, ast.nan_samples) == (bst.closest_rays, bst.any_rays, bst.any_occluded, bst.nan_samples)
    # and back to the stock integrators afterwards
    c, _, _ = ref.render(1, seed=11)
    d, _, _ = orc.render(1, seed=11)
    assert np.abs(c - d).max() <= 1e-9 * np.abs(c).max()


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_identical(seed):
    """Differential test on seeded random assemblies of every supported feature (scenes.random_scene)."""
    scene = S.random_scene(seed, 32, 24)
    ref = pyref.reference().scene(scene)
    orc = pyref.restatement("f64").scene(scene)
    for integ in (A.INTEGRATOR_MIS, A.INTEGRATOR_BSDF):
        a, an, ast = ref.render(3, integrator=integ, max_path_length=8, seed=seed, threads=4)
        b, bn, bst = orc.render(3, integrator=integ, max_path_length=8, seed=seed, threads=4)
        assert an == bn and a.max() > 0
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
        assert (ast.closest_rays, ast.any_rays, ast.any_occluded, ast.nan_samples) == (bst.closest_rays, bst.any_rays, bst.any_occluded, bst.nan_samples)
