"""Where do the fp32 and the fp64 render of the SAME samples part ways? Renders a crop fixture's tile (tests/golden/crop_*.npz:
the BASELINE configurations at the size they run) in both precisions with the decision log on (fcgpu_set_decision_log) and
prints the table of tests/common.py: divergence_table (the same function the GPU tests gate).
usage: divergence.py [fixture = headline] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as T

for name in (sys.argv[1:] or ["headline"]):
    z, scene, tile = T.load_crop(name)
    print(f"== crop_{name}: {z['builder']} {z['kwargs']}, tile {tile}, {int(z['spp'])} spp")
    img64, _, _, h64, f64 = T.render_crop(scene, z, tile, "f64", decision_log=True)
    img32, _, _, h32, f32 = T.render_crop(scene, z, tile, "f32", decision_log=True)
    T.divergence_table(h64, f64, img64, h32, f32, img32, tile[2] * tile[3])
