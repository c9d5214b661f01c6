"""bench.py end to end on small workloads: the JSON contract, the tile-sharded mode's in-bench bit-exact check, and the
N-rank paths (ranks started by bench.py itself) rehearsed with two ranks sharing the one GPU (--backend gloo)."""
import json
import os
import subprocess
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--width", "480", "--height", "270", "--tess", "8", "--steps", "2", "--warmup", "1", "--spp-per-step", "4", "--no-cpu-baseline", "--no-f64-leg"]


def run_bench(*args):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, *args], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout  # stdout carries the result line and nothing else
    return json.loads(lines[0])


def test_bench_line_single_gpu_tiles():
    d = run_bench("--shard", "tiles")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["unit"] == "Mrays/s" and d["value"] > 0
    assert d["samples_rendered"] == d["samples_expected"] == 480 * 270 * 4 * 2
    assert d["tile_check"]["bit_identical"] and d["tile_check"]["rows"] > 0
    # two rooflines of the dominant kernel: the bytes roof the contract names and the VALU roof that binds
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["unit"] == "GB/s"
    assert d["roofline_valu"]["bound"] == "valu" and 0 < d["roofline_valu"]["frac"] < 1 and d["roofline_valu"]["unit"] == "TFLOP/s"
    # counters of the PMC passes are printed only when they were taken on this build, and the line says which it is
    assert d["pmc_stale"] in (True, False, None) and (d["roofline"]["traffic"] is None) == (d["pmc_stale"] is not False)
    assert d["stack_overflows"] == 0 and d["nan_samples"] == 0


@pytest.mark.parametrize("shard", ["tiles", "samples"])
def test_bench_starts_its_own_ranks(shard):
    """`bench.py --gpus 2`: two ranks via torch.distributed.run, here sharing the GPU and reducing through the host."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--shard", shard)
    assert d["n_gpus"] == 2 and d["config"]["backend"] == "gloo" and d["config"]["shard"] == shard
    per_job = 480 * 270 * 4 * 2
    assert d["samples_rendered"] == d["samples_expected"] == (per_job if shard == "tiles" else 2 * per_job)
    assert d["scaling"] == ("strong" if shard == "tiles" else "weak")
    assert d["reduce_ms"] is not None and d["reduce_ms"] >= 0 and d["film_bytes"] == 24 * 480 * 270
    if shard == "tiles":
        # a rank takes the samples of `world` steps per pass, so that its batch is as large as a single GPU's
        assert d["config"]["paths_per_pass_per_rank"] == 480 * 135 * 4 * 2
        t = d["tile_check"]
        assert t["bit_identical"] and all(o == 1 for o in t["owners"])  # bands of the OTHER rank, re-rendered by rank 0
