"""The stratified mode of include/fcgpu_sampler.h (reference:
src/samplers/stratified_sampler.hpp:94-150). The header is the single source for the device, the
CPU restatement and the reference harness, so its properties are tested once, here, on the CPU;
film parity in this mode is in test_oracle_golden.py / test_gpu_parity.py (fixtures *_strat*)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import A, ROOT, S
from oracle import pyref


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("shim") / "sampler_shim.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so), os.path.join(ROOT, "tests", "cpp", "sampler_shim.c")], check=True)
    lib = C.CDLL(str(so))
    lib.shim_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.shim_random_bits.restype = C.c_uint64
    lib.shim_random_bits.argtypes = [C.c_uint64] + [C.c_uint32] * 4
    return lib


def draws(lib, seed, strat, px, py, dim, first, n):
    x, y = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    lib.shim_draws(seed, strat, px, py, dim, first, n, x.ctypes.data, y.ctypes.data)
    return x, y


def config(lib, strat):
    n, cnt, mul = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.shim_config(strat, C.byref(n), C.byref(cnt), C.byref(mul))
    return n.value, cnt.value, mul.value


@pytest.mark.parametrize("spp", [1, 4, 9, 10, 16, 64, 100, 1000, 1024, 4096])
def test_one_sample_per_cell_for_every_pixel_dimension_and_epoch(shim, spp):
    n, count, mul = config(shim, spp)
    assert n == int(np.floor(np.sqrt(spp))) and count == n * n  # stratified_sampler.hpp:15-16
    bounds = (np.arange(n + 1, dtype=np.uint64) * mul) >> np.uint64(8)  # cell j = [b(j), b(j+1)) in 2^-24 units
    assert bounds[-1] <= 1 << 24 and (1 << 24) - int(bounds[-1]) <= n   # the grid covers [0, 1) up to n ulps
    widths = np.diff(bounds.astype(np.int64))
    assert widths.min() >= widths.max() - 1
    for (px, py, dim, epoch) in [(0, 0, 0, 0), (13, 21, 5, 0), (1919, 1079, 61, 2), (5, 5, 1, 7)]:
        x, y = draws(shim, 3, spp, px, py, dim, epoch * count, count)
        assert x.max() < 1 << 24 and y.max() < 1 << 24
        j = np.searchsorted(bounds, x, side="right") - 1
        i = np.searchsorted(bounds, y, side="right") - 1
        assert np.array_equal(np.sort(i * n + j), np.arange(count))  # a permutation of all cells


def test_order_and_jitter_vary_with_pixel_dimension_epoch_and_seed(shim):
    base = draws(shim, 3, 64, 10, 10, 2, 0, 64)
    for other in (draws(shim, 3, 64, 11, 10, 2, 0, 64), draws(shim, 3, 64, 10, 10, 3, 0, 64), draws(shim, 3, 64, 10, 10, 2, 64, 64),
                  draws(shim, 4, 64, 10, 10, 2, 0, 64)):
        assert (base[0] != other[0]).mean() > 0.9 and (base[1] != other[1]).mean() > 0.9
    # the visiting order of the cells is not the raster order and differs between dimensions
    n = 8
    cells = lambda d: (d[1] >> 21) * n + (d[0] >> 21)
    assert not np.array_equal(cells(base), np.arange(64))
    assert not np.array_equal(cells(base), cells(draws(shim, 3, 64, 10, 10, 3, 0, 64)))


def test_jitter_is_uniform_inside_the_cells(shim):
    # pooled over many pixels, the in-cell offsets fill [0, 1) evenly (chi-square over 16 bins, 2 axes)
    n, count, mul = config(shim, 16)
    off = []
    for px in range(200):
        x, y = draws(shim, 9, 16, px, 3, 4, 0, count)
        off.append((x % (1 << 22)) / float(1 << 22))
        off.append((y % (1 << 22)) / float(1 << 22))
    h, _ = np.histogram(np.concatenate(off), bins=16, range=(0, 1))
    chi2 = ((h - h.mean()) ** 2 / h.mean()).sum()
    assert chi2 < 45  # 15 dof: p ~ 1e-4


def test_random_mode_is_the_original_stream(shim):
    for seed, px, py, s, dim in [(0, 0, 0, 0, 0), (7, 1919, 1079, 1023, 55), (123456789, 17, 4000, 4095, 3)]:
        x, y = draws(shim, seed, 0, px, py, dim, s, 1)
        bits = shim.shim_random_bits(seed, px, py, s, dim)
        assert int(x[0]) == bits >> 40 and int(y[0]) == (bits & 0xFFFFFFFF) >> 8


@pytest.mark.skipif(not pyref.have_reference(), reason="oracle/_ref/libfcref.so not built")
def test_restatement_equals_reference_in_stratified_mode():
    sc = S.primitives_zoo(32, 32)
    a = pyref.reference().scene(sc).render(9, seed=2, stratified=9, threads=4)
    b = pyref.restatement("f64").scene(sc).render(9, seed=2, stratified=9, threads=4)
    assert a[1] == b[1] and np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(a[0]).max()
    assert (a[2].closest_rays, a[2].any_rays) == (b[2].closest_rays, b[2].any_rays)


def test_stratified_lowers_the_error_at_equal_sample_count():
    """What the reference's stratified sampler is for: at 16 spp the stratified estimate is closer
    to the converged image than the random one (direct light + one bounce, where dimensions 0-7 matter most)."""
    sc = S.room_planes(40, 24)
    o = pyref.restatement("f64").scene(sc)
    ref = o.render(1024, max_path_length=3, seed=100, threads=8)
    ref = ref[0] / ref[1]
    err = {}
    for mode in (None, 16):
        e = []
        for seed in range(4):
            r = o.render(16, max_path_length=3, seed=seed, stratified=mode, threads=8)
            e.append(np.mean((r[0] / r[1] - ref) ** 2))
        err[mode] = float(np.mean(e))
    print("MSE random", err[None], "stratified", err[16])
    assert err[16] < 0.8 * err[None]
