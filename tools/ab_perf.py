"""Same-box A/B timing of two builds of libfcgpu.so (development tool).

usage: python tools/ab_perf.py [--reps N] [--spp N] [--tess N] LIB_A LIB_B ...
Each measurement runs in a child process (FCGPU_LIBRARY selects the build); builds alternate so that
box-to-box and thermal drift cancel. Prints per-stage device ms (HIP events inside the library).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import importlib, sys, time, os
sys.path.insert(0, %r)
S = importlib.import_module("ray-tracing_amd.scenes")
Rn = importlib.import_module("ray-tracing_amd.renderer")
spp, tess = int(sys.argv[1]), int(sys.argv[2])
sc = S.room(1920, 1080, tess=tess)
ctx = Rn.GpuContext(0)
ctx.upload_scene(sc)
ctx.render_pass(0, 2); ctx.sync()
ctx.set_profiling(True); ctx.stage_times(); ctx.reset_counters(); ctx.sync()
t0 = time.time(); ctx.render_pass(2, spp); ctx.sync(); dt = time.time() - t0
st = ctx.stage_times(); c = ctx.counters()
rays = c.closest_rays + c.shadow_rays
rgb, n = ctx.read_film()
print("RESULT %%.1f Msamples/s %%.3f Grays/s | " %% (1920*1080*spp/dt/1e6, rays/dt/1e9) +
      " ".join("%%s %%.1f" %% (k, v[0]) for k, v in st.items()) + " | mean %%.6f" %% (rgb / n).mean())
""" % ROOT


def main():
    # usage: ab_perf.py [--reps N] [--spp N] [--tess N] LIB...
    args = sys.argv[1:]
    opt = {"--reps": "2", "--spp": "16", "--tess": "64"}
    libs = []
    while args:
        a = args.pop(0)
        if a in opt:
            opt[a] = args.pop(0)
        else:
            libs.append(os.path.abspath(a))
    for r in range(int(opt["--reps"])):
        for tag, lib in zip("ABCDEFGH", libs):
            env = dict(os.environ, FCGPU_LIBRARY=lib)
            out = subprocess.run([sys.executable, "-c", CHILD, opt["--spp"], opt["--tess"]], env=env, capture_output=True,
                                 text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            print(tag, os.path.basename(lib), line[0] if line else ("FAILED " + out.stderr[-400:]), flush=True)
            if not line:
                sys.exit(1)


if __name__ == "__main__":
    main()
