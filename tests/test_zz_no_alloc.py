"""fcgpu_trace_closest_noalloc (the entry point for callers that cannot allocate: scratch in a static device array)
against fcgpu_trace_closest_device, which it must equal bit for bit: same kernel, other scratch. Two calls in a row per
scene, the second with a smaller input; a shallow hierarchy (no spill columns in use) and a deep one (stack need 47 of
the 62 the variant supports, with rays that really fill it: the spill columns of the static array are exercised); the limits are refused."""
import numpy as np
import pytest

from common import A, R, S, cards_rays, cards_scene, ribbon_scene

pytestmark = pytest.mark.gpu


def _rays(kind, n, seed):
    if kind == "cards":  # rays that fill the traversal stack to the hierarchy's worst case (common.cards_scene)
        return cards_rays(n, seed)
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=A.ray_dtype)
    rays["origin"] = rng.uniform([-9, 0.5, -8], [9, 7, 15], size=(n, 3))
    d = rng.normal(size=(n, 3))
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["t_max"] = np.inf
    return rays


@pytest.mark.parametrize("kind,precision", [("room", "f32"), ("cards", "f64"), ("cards", "f32")])
def test_noalloc_trace_equals_the_allocating_entry_point(kind, precision):
    import torch

    scene = S.room(64, 36, tess=16) if kind == "room" else cards_scene(120)
    dev = torch.device("cuda", 0)
    with R.GpuContext(0) as ctx:
        ctx.set_options(precision=precision)
        ctx.upload_scene(scene)
        depth = int(ctx.counters().reserved[1])
        assert (depth == 47) if kind == "cards" else (depth <= 28)
        for call, n in enumerate((50000, 7001)):  # two calls in a row, the second smaller
            rays = _rays(kind, n, seed=10 + call)
            d_rays = torch.from_numpy(rays.view(np.uint8).copy()).to(dev)
            want = torch.full((n * A.hit_dtype.itemsize,), 0xAB, dtype=torch.uint8, device=dev)
            got = torch.full(((n + 64) * A.hit_dtype.itemsize,), 0xCD, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.trace_closest_device(n, d_rays.data_ptr(), want.data_ptr())
            ctx.sync()
            spilled_before = int(ctx.counters().reserved[5])
            ctx.trace_closest_noalloc(n, d_rays.data_ptr(), got.data_ptr())
            ctx.sync()
            c = ctx.counters()
            assert c.stack_overflows == 0
            if kind == "cards":
                assert int(c.reserved[5]) > spilled_before, "no push reached the static array's spill columns"
            a, b = want.cpu().numpy(), got.cpu().numpy()
            assert np.array_equal(a, b[: len(a)])                    # bit for bit, ids, distances and barycentrics
            assert (b[len(a):] == 0xCD).all()                        # nothing written past the n hits
            hits = b[: len(a)].view(A.hit_dtype)
            assert (hits["prim"] != 0xFFFFFFFF).mean() > 0.2
        # limits are refused with a clear error, nothing is launched
        with pytest.raises(R.FcgpuError) as e:
            ctx.trace_closest_noalloc((1 << 30) + 1, d_rays.data_ptr(), got.data_ptr())
        assert e.value.code == A.E_LIMIT and "FCGPU_NOALLOC_MAX_RAYS" in str(e.value)


def test_noalloc_trace_refuses_a_hierarchy_beyond_its_scratch():
    import torch

    with R.GpuContext(0) as ctx:
        ctx.set_options(precision="f64")
        ctx.upload_scene(ribbon_scene(420))  # stack need ~209 > 62
        assert int(ctx.counters().reserved[1]) > 62
        buf = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        with pytest.raises(R.FcgpuError) as e:
            ctx.trace_closest_noalloc(8, buf.data_ptr(), buf.data_ptr())
        assert e.value.code == A.E_LIMIT and "FCGPU_NOALLOC_MAX_STACK" in str(e.value)
