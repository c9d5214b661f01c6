"""Development tool: is a pixel that differs between the device's fp64 film and the restatement's WRONG or ILL-CONDITIONED?
Renders a random scene's pixel with the CPU restatement while every transcendental result is moved by -1 / 0 / +1 ulp (the
distance between the device's libm and glibc; oracle/fc_oracle.cpp: lm::nudge) for several noise seeds and prints the spread.
usage: ulp_sensitivity.py SEED X Y [SPP = 3] [MAX_PATH_LENGTH = 8]   (the sizes of tools/fuzz_random_scenes.py)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
S = importlib.import_module("ray-tracing_amd.scenes")
from oracle import pyref
seed, x, y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 3
mpl = int(sys.argv[5]) if len(sys.argv) > 5 else 8
sc = S.random_scene(seed, 32, 24)
o = pyref.restatement("f64")
scene = o.scene(sc)  # transforms are set up with the plain libm, as the device's host code does
vals = []
for noise in range(0, 17):
    o.set_ulp_noise(noise)
    rgb, n, st = scene.render(spp, max_path_length=mpl, seed=seed, threads=8)
    vals.append(rgb[y, x].copy())
o.set_ulp_noise(0)
vals = np.array(vals)
base = vals[0]
rel = np.abs(vals[1:] - base).max(1) / max(np.abs(base).max(), 1e-300)
print("pixel", (x, y), "plain libm:", base)
print("relative change under +-1 ulp noise, 16 seeds: max %.3e, median %.3e" % (rel.max(), np.median(rel)))
