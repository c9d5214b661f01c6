"""Runs the A/B child of ab_perf.py under several environment settings (development tool).
usage: python tools/knob_sweep.py [--spp N] "A=1 B=2" "A=3" ...   (each argument = one configuration; "" = defaults)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_perf
args = sys.argv[1:]
spp = "16"
if args and args[0] == "--spp":
    spp = args[1]
    args = args[2:]
for cfg in args:
    env = dict(os.environ)
    for kv in cfg.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, "-c", ab_perf.CHILD, spp, "64"], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(f"[{cfg}]", line[0] if line else "FAILED " + out.stderr[-300:], flush=True)
