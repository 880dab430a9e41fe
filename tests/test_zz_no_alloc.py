"""The one entry point that allocates nothing (pm_emit_range_static) against the entry point it is a variant of
(pm_emit_range), through the C-ABI, bit for bit, on two calls in a row, the second with a smaller input; and its limit.

Collected last on purpose (the round's end runs `pytest -x`)."""
import numpy as np
import pytest
import torch  # before the library: one HIP runtime per process (raytracer_amd/capi.py: load_library)

pytestmark = pytest.mark.gpu


def _pair(ctx, kind, seed, begin, count, cap):
    import torch
    import raytracer_amd as ra
    lc = (0.6, 0.6, 0.6)
    # both outputs pre-filled with the same pattern: records beyond the count must be left alone by both
    a = torch.full((cap, 8), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    b = torch.full((cap, 8), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    n_ref = ctx.emit_range(kind, seed, lc, begin, count, a.data_ptr(), cap)
    n_new = ctx.emit_range_static(kind, seed, lc, begin, count, b.data_ptr(), cap)
    torch.cuda.synchronize()
    assert n_new == n_ref
    assert 0 < n_ref <= count
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    return n_ref, b.cpu().numpy()


@pytest.mark.parametrize("kind", [0, 1])
def test_emit_range_static_equals_emit_range_twice(gpu_ctx, orc, kind):
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    big = ra.capi.PM_EMIT_STATIC_MAX_ATTEMPTS          # the largest call the static scratch takes
    n1, rec1 = _pair(gpu_ctx, kind, 4242, 1000, big, big)
    # the second call is smaller and starts elsewhere: whatever the first left in the scratch must not show
    n2, rec2 = _pair(gpu_ctx, kind, 4242, 777777, 5000, 4000)
    assert n2 < n1
    # and a cap below the count: only the first cap records are written, the count is still the range's
    n3, rec3 = _pair(gpu_ctx, kind, 4242, 777777, 5000, 100)
    assert n3 == n2 and np.array_equal(rec3[:100], rec2[:100])
    # the records are the oracle's for the same attempts (the first n of a map are attempts 0..): spot check from attempt 0
    n4, rec4 = _pair(gpu_ctx, kind, 4242, 0, 3000, 3000)
    want, _ = orc.emit(orc.default_scene(), kind, 4242, n4, trig=orc.TRIG_DET)
    assert rec4[:n4].tobytes() == want.tobytes()


def test_emit_range_static_through_the_sphere_grid(gpu_ctx):
    """10 000 spheres: the emission is the stream kernel (pm_trace.h), whose work counter lives in the scratch too."""
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.many_spheres_scene())
    try:
        n1, _ = _pair(gpu_ctx, 0, 99, 0, ra.capi.PM_EMIT_STATIC_MAX_ATTEMPTS, ra.capi.PM_EMIT_STATIC_MAX_ATTEMPTS)
        n2, _ = _pair(gpu_ctx, 1, 99, 31, 3000, 3000)
        assert n2 < n1
    finally:
        gpu_ctx.upload_scene(ra.default_scene())


def test_emit_range_static_rejects_what_the_scratch_cannot_hold(gpu_ctx):
    import torch
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    out = torch.zeros((16, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(ra.PMError, match="at most 131072 attempts"):
        gpu_ctx.emit_range_static(0, 1, (0.6, 0.6, 0.6), 0, ra.capi.PM_EMIT_STATIC_MAX_ATTEMPTS + 1, out.data_ptr(), 16)
    assert gpu_ctx.emit_range_static(0, 1, (0.6, 0.6, 0.6), 0, 0, out.data_ptr(), 16) == 0
