import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from common import A, R, S
from oracle import pyref
from make_golden import bsdf_queries
q = bsdf_queries(20000, 9)
m = A.Material(kind=A.MATERIAL_GLASS)
m.color_a[:] = (1, 1, 1); m.color_b[:] = (1, 1, 1); m.roughness[:] = (0, 0)
o = pyref.restatement("f64")
np.set_printoptions(precision=7, linewidth=200)
with R.GpuContext(0) as ctx:
    ctx.set_options(precision="f32")
    for ea, eb in ((1.0, 1.5), (1.5, 1.0)):
        ref = o.bsdf_table(m, q, eta_a=ea, eta_b=eb)
        got = ctx.bsdf_table(m, q, eta_a=ea, eta_b=eb)
        print(ea, eb, "s_ok mismatches", (ref["s_ok"] != got["s_ok"]).sum())
        both = (ref["s_ok"] == 1) & (got["s_ok"] == 1)
        d = np.abs(ref["s_wi"] - got["s_wi"]).max(axis=1)
        bad = both & (d > 1e-4)
        print("  s_wi differs > 1e-4 on", bad.sum(), "of", both.sum())
        for k in np.flatnonzero(bad)[:6]:
            print("   wo", q["wo"][k], "u1", q["u1"][k], "gpu", got["s_wi"][k], got["s_value"][k][0], got["s_pdf"][k], "ref", ref["s_wi"][k], ref["s_value"][k][0], ref["s_pdf"][k])
        dv = np.abs(ref["s_value"] - got["s_value"]).max(axis=1) / np.maximum(np.abs(ref["s_value"]).max(axis=1), 1e-6)
        print("  s_value rel diff > 1e-3 on", (both & (dv > 1e-3)).sum())
