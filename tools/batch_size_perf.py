"""Development tool: throughput against the batch size (paths in flight). usage: python tools/batch_size_perf.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("ray-tracing_amd.scenes")
Rn = importlib.import_module("ray-tracing_amd.renderer")
sc = S.room(1920, 1080, tess=64)
for log2 in (24, 25, 26, 27):
    with Rn.GpuContext(0) as ctx:
        ctx.set_options(max_paths_in_flight=1 << log2)
        ctx.upload_scene(sc)
        ctx.render_pass(0, 2); ctx.sync()
        t0 = time.time(); ctx.render_pass(2, 128); ctx.sync(); dt = time.time() - t0
        print(f"2^{log2} paths in flight: {1920 * 1080 * 128 / dt / 1e6:.1f} Msamples/s", flush=True)
