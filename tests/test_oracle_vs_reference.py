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
