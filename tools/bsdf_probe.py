"""Development tool: rows of tests/golden/bsdf_tables.npz where the device differs from the reference."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import R
from test_gpu_parity import _bsdf_table_cases
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
np.set_printoptions(precision=6, linewidth=200)
with R.GpuContext(0) as ctx:
    ctx.set_options(precision=prec)
    shown = 0
    for name, ei, fi, m, ea, eb, fr, ref, q in _bsdf_table_cases():
        got = ctx.bsdf_table(m, q, ng=fr[0], tangent=fr[1], ns=fr[2] / np.linalg.norm(fr[2]), bitangent=fr[3], eta_a=ea, eta_b=eb)
        for field in ("f", "pdf", "s_wi", "s_value", "s_pdf"):
            a, b = got[field].reshape(len(q), -1), ref[field].reshape(len(q), -1)
            bad = ~np.isclose(a, b, rtol=tol, atol=1e-300, equal_nan=True).all(axis=1)
            if bad.any():
                i = np.flatnonzero(bad)
                print(f"{name} eta#{ei} ({ea},{eb}) frame {fi} {field}: {len(i)} rows differ; lobes {np.unique(ref['lobe'][i])} s_ok ref {np.unique(ref['s_ok'][i])} gpu {np.unique(got['s_ok'][i])}")
                for k in i[:3]:
                    print("   row", k, "gpu", a[k], "ref", b[k], "wo.y", q["wo"][k][1], "wi.y", q["wi"][k][1], "gpu s_wi", got["s_wi"][k], "ref s_wi", ref["s_wi"][k], "pdf", got["s_pdf"][k], ref["s_pdf"][k])
                shown += 1
print("tables with differences:", shown)
