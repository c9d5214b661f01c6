"""Host-side logic that needs no GPU: scene descriptions, tile sharding, sampler statistics."""
import os
import ctypes as C

import numpy as np
import pytest

from common import A, R, S, cards_scene, ribbon_scene
from oracle import pyref


def test_room_primitive_counts():
    sc = S.room(width=64, height=36, tess=4)
    counts = sc.primitive_counts()
    assert counts[0] == 12 * 4 * 4 and counts[1] == 8 * 10 and counts[2] == 8 * 2 and counts[3:] == [1, 1, 1]
    d = sc.desc()
    assert d.entity_count == 6 and d.medium_count == 1 and d.index_total == 3 * sum(counts[:3])


def test_shard_tiles_partition_every_pixel_once():
    for (w, h, world, tile) in ((1920, 1080, 8, 64), (7680, 4320, 8, 128), (100, 37, 3, 16), (64, 64, 2, 64)):
        cover = np.zeros((h, w), np.int32)
        sizes = []
        for rank in range(world):
            tiles = R.shard_tiles(w, h, rank, world, tile)
            sizes.append(sum(t[2] * t[3] for t in tiles))
            for x0, y0, tw, th in tiles:
                cover[y0:y0 + th, x0:x0 + tw] += 1
        assert cover.min() == 1 and cover.max() == 1
        if w * h > 100000:
            assert max(sizes) <= 1.15 * (w * h / world)  # balanced within 15 %


def test_sampler_is_uniform_and_decorrelated():
    # draws of the oracle build of fcgpu_sampler.h through a 1-lobe table query would be indirect;
    # re-implement the integer mixing with numpy uint64 (wraps mod 2^64) and test statistics
    def mix64(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        px = np.arange(256, dtype=np.uint64)[:, None]
        s = np.arange(256, dtype=np.uint64)[None, :]
        key = mix64(mix64(px | (np.uint64(3) << np.uint64(20)) | (s << np.uint64(40))) ^ np.uint64(0xD1B54A32D192ED03))
        bits = mix64(key + np.uint64(1) * np.uint64(0x9E3779B97F4A7C15))
    x = (bits >> np.uint64(40)).astype(np.float64) / 2 ** 24
    y = ((bits & np.uint64(0xFFFFFFFF)) >> np.uint64(8)).astype(np.float64) / 2 ** 24
    assert abs(x.mean() - 0.5) < 0.005 and abs(y.mean() - 0.5) < 0.005
    assert abs(np.corrcoef(x.ravel(), y.ravel())[0, 1]) < 0.02
    assert abs(np.corrcoef(x[:, :-1].ravel(), x[:, 1:].ravel())[0, 1]) < 0.02
    hist, _ = np.histogram(x, bins=16, range=(0, 1))
    assert hist.min() > 0.9 * x.size / 16


def test_oracle_tile_invariance():
    """Rendering a tile equals the same pixels of the full frame (the sampler is keyed by pixel)."""
    sc = S.room_planes(width=32, height=18)
    o = pyref.restatement("f64").scene(sc)
    full, _, _ = o.render(2, seed=1)
    part, _, _ = o.render(2, seed=1, tile=(5, 3, 12, 9))
    assert np.array_equal(full[3:12, 5:17], part[3:12, 5:17])
    a, _, _ = o.render(1, seed=1, sample_begin=0)
    b, _, _ = o.render(1, seed=1, sample_begin=1)
    assert np.allclose(a + b, full, rtol=1e-14)


def test_fast_div_matches_hardware_divide(tmp_path):
    """csrc/fc_types.h fast_div (slot -> pixel mapping on the device) against n / d on the CPU."""
    import subprocess
    exe = tmp_path / "test_fastdiv"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_fastdiv.cpp")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", str(exe), src], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_check_scene_builds_the_hierarchy_without_a_device():
    """fcgpu_check_scene: validation + host BVH build, no GPU. The mask stand-in at detail 256 has centroid ranges of a
    few ulps on the minor axes, where the float partition point of a bucket split can leave one side empty: the
    builder has to fall back to the median there instead of recursing for ever."""
    sc = S.room(width=64, height=36, tess=16)
    for builder in ("host", "reference"):
        st = R.check_scene(sc, builder)
        assert st.primitives == sc.primitive_total() and st.triangles == sc.primitive_total() - 3  # three spheres
        assert st.wide_nodes * 4 >= st.primitives / 2 and 0 < st.max_stack <= 30
        assert st.lobe_mask & 1  # lambertian walls
    big = S.mask_scene(width=60, height=90, detail=256, texture_size=16)
    a, b = R.check_scene(big, "host"), R.check_scene(big, "reference")
    assert a.primitives == b.primitives == big.primitive_total() and a.binary_nodes == b.binary_nodes
    bad = S.room_planes(32, 18)
    bad.entities[0].material = 99
    with pytest.raises(R.FcgpuError) as e:
        R.check_scene(bad)
    assert e.value.code == A.E_INVALID and "entity" in str(e.value)
    lib = R.load_library()
    assert lib.fcgpu_check_scene(None, A.BVH_HOST_SAH, None) == A.E_INVALID
    desc = sc.desc()
    assert lib.fcgpu_check_scene(C.byref(desc), A.BVH_DEVICE_LBVH, None) == A.E_INVALID


def test_threaded_bvh_build_gives_the_sequential_tree(monkeypatch):
    """Above 65 536 primitives the host build runs ranges of the tree on several threads and splices them into
    depth-first order: node counts and the worst-case stack must not depend on the thread count."""
    sc = S.room(width=64, height=36, tess=80)  # 76 899 primitives
    stats = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("FCGPU_BUILD_THREADS", threads)
        st = R.check_scene(sc, "host")
        stats.append((st.primitives, st.binary_nodes, st.wide_nodes, st.max_stack))
    assert stats[0] == stats[1] == stats[2] and stats[0][0] == sc.primitive_total()


def test_every_scene_builder_flattens_on_the_host():
    """All stand-in scenes go through validation, the BVH build and the per-triangle shading records without a device
    (scenes with a single mesh-less placeholder triangle record included)."""
    for name, build in S.BUILDERS.items():
        sc = S.random_scene(1, 32, 24) if name == "random_scene" else build(width=32, height=24)
        st = R.check_scene(sc)
        assert st.primitives == sc.primitive_total(), name


def test_every_scene_flattens_eight_wide_too():
    """fcgpu_options.bvh_width = 8 (Node8: octant-ordered child slots, consecutive interior children, consecutive leaf
    records): every stand-in scene flattens on the host, with fewer nodes than 4-wide and the few group entries its
    traversal stack needs; a chain-shaped hierarchy deeper than the 15 entries is reported, not built wrong."""
    for name, build in S.BUILDERS.items():
        sc = S.random_scene(1, 32, 24) if name == "random_scene" else build(width=32, height=24)
        four, eight = R.check_scene(sc), R.check_scene(sc, bvh_width=8)
        assert eight.primitives == four.primitives == sc.primitive_total(), name
        assert 0 < eight.wide_nodes <= four.wide_nodes and eight.max_stack <= four.max_stack, name
    room = S.room(width=64, height=36, tess=64)
    four, eight = R.check_scene(room), R.check_scene(room, bvh_width=8)
    assert (four.wide_nodes, four.max_stack) == (12112, 28) and (eight.wide_nodes, eight.max_stack) == (8459, 5)
    assert R.check_scene(ribbon_scene(420), bvh_width=8).max_stack > 15  # the upload refuses it (FCGPU_E_LIMIT): GPU test


def test_card_scenes_straddle_the_lds_stack_capacity():
    """The GPU tests of the 28 .. 32-entry window (test_bvh_whose_stack_need_is_at_the_lds_capacity) and of the
    no-allocation entry point (47 of its 62 entries) rely on these depths; fcgpu_check_scene builds the same hierarchy."""
    for n, depth in ((58, 28), (60, 29), (66, 30), (67, 31), (72, 32), (120, 47)):
        assert R.check_scene(cards_scene(n)).max_stack == depth
    assert R.check_scene(ribbon_scene(420)).max_stack > 62


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`bench.py --gpus N` starts N ranks itself; where fewer GPUs are visible (none in the CPU container) it must fail
    loudly instead of printing a line with n_gpus smaller than asked."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr and "metric" not in p.stdout
