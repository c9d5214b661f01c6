import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import A, R, cards_scene, cards_rays
from oracle import pyref
for n in (60, 64, 68, 72, 120, 155):
    sc = cards_scene(n)
    rays = cards_rays(4000, 3)
    ref = pyref.restatement("f64").scene(sc).trace_closest(rays)
    for prec in ("f64", "f32"):
        with R.GpuContext(0) as ctx:
            ctx.set_options(precision=prec)
            ctx.upload_scene(sc)
            depth = int(ctx.counters().reserved[1])
            ctx.reset_counters()
            h = ctx.trace_closest(rays)
            c = ctx.counters()
            print(n, prec, "depth", depth, "spilled", int(c.reserved[5]), "lost", int(c.stack_overflows), "prim mismatches", int((h["prim"] != ref["prim"]).sum()),
                  "backstop hits", int((ref["prim"] >= n).sum()), "miss", int((ref["prim"] == 0xFFFFFFFF).sum()))
