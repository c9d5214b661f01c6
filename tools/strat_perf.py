"""Throughput of the stratified sampler mode against the random stream (development tool)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("ray-tracing_amd.scenes")
R = importlib.import_module("ray-tracing_amd.renderer")
sc = S.room(1920, 1080, tess=64)
for mode in (None, 64, None, 64):
    ctx = R.GpuContext(0)
    ctx.upload_scene(sc)
    ctx.set_sampler(0, stratified=mode)
    ctx.render_pass(0, 16); ctx.sync()
    t0 = time.time(); ctx.render_pass(16, 48); ctx.sync(); dt = time.time() - t0
    rgb, n = ctx.read_film()
    print("stratified" if mode else "random    ", f"{1920*1080*48/dt/1e6:.1f} Msamples/s  mean {(rgb/n).mean():.5f}", flush=True)
    ctx.close()
