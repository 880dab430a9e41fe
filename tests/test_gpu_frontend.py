"""GPU tests of the entry points either side of the frame: pm_intersect_rays (SceneObject::intersect / Scene::intersect
for caller-given rays: the reference's 3 000 intersection KATs replayed on gfx950), pm_downsample / pm_render_image (the
image front-end of src/Image.cpp:36-57 on the device), and the error paths a frame can end in."""
import ctypes
import os

import numpy as np
import pytest
import torch  # before the library: one HIP runtime per process

from .conftest import golden

pytestmark = pytest.mark.gpu


def test_reference_intersection_kats_on_the_gpu(orc):
    """tests/golden/kat_intersect.bin (written by the compiled reference: inside-sphere origins, mul == 0, d.a == 0 with its
    inf / NaN paths, un-normalised directions, a moving sphere): every record through prim_hit / prim_normal on the GPU."""
    import raytracer_amd as ra
    kat = golden("kat_intersect.bin", np.float64).reshape(-1, 15)
    prims = np.concatenate([ra.default_scene(), golden("kat_moving_sphere.bin", ra.prim_dtype)[:1]])
    ctx = ra.Context(0)
    ctx.upload_scene(prims)
    res0 = np.full((len(kat), 3), 7.25); nrm0 = np.full((len(kat), 3), -3.5)
    hit, res, nrm = ctx.intersect_rays(kat[:, 1:4], kat[:, 4:7], kat[:, 7], prim=kat[:, 0].astype(np.int32), res=res0.copy(), nrm=nrm0.copy())
    want = kat[:, 8] != 0
    assert len(kat) == 3000 and np.array_equal(hit != 0, want) and 100 < want.sum() < len(kat)
    assert np.array_equal(res[want].view(np.uint64), kat[want, 9:12].view(np.uint64))
    assert np.array_equal(nrm[want].view(np.uint64), kat[want, 12:15].view(np.uint64))
    # a miss leaves the caller's result and normal alone, like the reference's out-parameters
    assert np.array_equal(res[~want], res0[~want]) and np.array_equal(nrm[~want], nrm0[~want])
    # the same rays against the whole scene, with and without an excluded object: Scene::intersect of the oracle
    rng = np.random.default_rng(5)
    sub = rng.choice(len(kat), 600, replace=False)
    excl = rng.integers(-1, len(prims), size=len(sub)).astype(np.int32)
    hit, res, nrm = ctx.intersect_rays(kat[sub, 1:4], kat[sub, 4:7], kat[sub, 7], exclude=excl)
    for j, i in enumerate(sub):
        obj, r, n = orc.scene_intersect(prims, kat[i, 1:4], kat[i, 4:7], int(excl[j]), kat[i, 7])
        assert hit[j] == obj
        if obj >= 0:
            assert np.array_equal(res[j].view(np.uint64), r.view(np.uint64)) and np.array_equal(nrm[j].view(np.uint64), n.view(np.uint64))
    with pytest.raises(ra.PMError, match="not a primitive"):
        ctx.intersect_rays(kat[:2, 1:4], kat[:2, 4:7], kat[:2, 7], prim=np.array([0, 8], dtype=np.int32))
    ctx.close()


@pytest.mark.parametrize("grid", ["1", "0"])
def test_scene_intersect_on_a_large_scene(orc, grid, monkeypatch):
    """Arbitrary rays through the sphere grid (and the linear scan) of the 400-primitive moving / mirror / glass scene."""
    import raytracer_amd as ra
    prims = golden("scene_moving100.bin", ra.prim_dtype)
    monkeypatch.setenv("PM_SCENE_GRID", grid)
    ctx = ra.Context(0)
    ctx.upload_scene(prims)
    rng = np.random.default_rng(11)
    n = 1500
    src = rng.uniform([-4.9, -4.9, -11], [4.9, 4.9, 9.9], size=(n, 3))
    d = rng.normal(size=(n, 3)); d /= np.sqrt((d * d).sum(1))[:, None]
    d[::7] *= 2.5                                             # not every direction is a unit vector
    d[::11, 0] = 0.0; d[::13, 1] = 0.0                        # axis-parallel components: the planes' division by zero
    dt = rng.uniform(0, 1, size=n)
    excl = rng.integers(-1, len(prims), size=n).astype(np.int32)
    hit, res, nrm = ctx.intersect_rays(src, d, dt, exclude=excl)
    for i in range(n):
        obj, r, nn = orc.scene_intersect(prims, src[i], d[i], int(excl[i]), dt[i])
        assert hit[i] == obj, i
        if obj >= 0:
            assert np.array_equal(res[i].view(np.uint64), r.view(np.uint64)) and np.array_equal(nrm[i].view(np.uint64), nn.view(np.uint64))
    assert (hit >= 0).sum() > n // 2
    ctx.close()


def box_filter(pane, s, flip=True):
    h, w = pane.shape[0] // s, pane.shape[1] // s
    ch = [((pane >> sh) & 0xFF).astype(np.int64).reshape(h, s, w, s).sum(axis=(1, 3)) // (s * s) for sh in (16, 8, 0)]
    out = (0xFF << 24) | (ch[0] << 16) | (ch[1] << 8) | ch[2]
    return (out[::-1] if flip else out).astype(np.uint32)


@pytest.mark.parametrize("s", [1, 2, 3, 5])
def test_downsample_is_the_references_box_filter(gpu_ctx, s):
    rng = np.random.default_rng(s)
    w, h = 67, 45                                             # not multiples of the launch tile
    pane = rng.integers(0, 1 << 32, size=(h * s, w * s), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(gpu_ctx.downsample(pane, s), box_filter(pane, s))
    assert np.array_equal(gpu_ctx.downsample(pane, s, flip=False), box_filter(pane, s, flip=False))
    d_in = torch.from_numpy(pane.view(np.int32)).cuda(); d_out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    gpu_ctx.downsample_device(d_in.data_ptr(), w, h, s, d_out.data_ptr())
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), box_filter(pane, s))
    out = np.zeros((h, w), dtype=np.uint32)
    assert gpu_ctx.lib.pm_downsample(gpu_ctx.handle, pane.ctypes.data, 0, w, h, 0, 1, out.ctypes.data, 0) != 0   # a zero sample ratio is refused


def test_render_image_equals_render_plus_box_filter(gpu_ctx, orc):
    """pm_render_image (pane kept on the device) against pm_render to the host + the reference's filter, at a small size
    against the oracle's pane, and at BASELINE config 3's (3840 x 2160, 4 spp: pane 7680 x 4320) against pm_render's."""
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    n, seed = 4096, 9
    g = gpu_ctx.create_map(0, n, seed); c = gpu_ctx.create_map(1, n, seed)
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    for w, h, s in ((160, 90, 2), (96, 54, 3)):
        img, st = gpu_ctx.render_image(g, c, gpu_ctx.params(w * s, h * s, seed), s)
        want, _ = orc.render(orc.default_scene(), og, oc, orc.params(w * s, h * s, seed), nthreads=8)
        assert np.array_equal(img, box_filter(want, s))
    og.close(); oc.close()
    w, h, s = 3840, 2160, 2
    img, st = gpu_ctx.render_image(g, c, gpu_ctx.params(w * s, h * s, seed), s)
    pane, _ = gpu_ctx.render(g, c, gpu_ctx.params(w * s, h * s, seed))
    assert np.array_equal(img, box_filter(pane, s))
    assert st.closest_hit_queries >= w * s * h * s
    g.destroy(); c.destroy()


def test_a_frame_that_overflows_its_queue_is_reported(gpu_ctx, monkeypatch):
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    g = gpu_ctx.create_map(0, 2048, 1); c = gpu_ctx.create_map(1, 2048, 1)
    rp = gpu_ctx.params(320, 200, 1)
    good, _ = gpu_ctx.render(g, c, rp)
    gpu_ctx.set_option("sp_cap_limit", 1000)                  # far fewer slots than the frame has shade points
    with pytest.raises(ra.PMError, match="shade point queue overflowed"):
        gpu_ctx.render(g, c, rp)
    gpu_ctx.set_option("sp_cap_limit", 0)
    again, _ = gpu_ctx.render(g, c, rp)                       # and the context is fine afterwards
    assert np.array_equal(good, again)
    g.destroy(); c.destroy()


def test_an_overflowing_frame_with_spread_out_tiles(monkeypatch):
    """The same on the 10 000 sphere scene with the deep tiles listed for the spatial sort (PM_LOOSE_TILES): a queue that is
    too small must not send the kernels behind it out of bounds through that list."""
    import raytracer_amd as ra
    monkeypatch.setenv("PM_LOOSE_TILES", "2")
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(ra.many_spheres_scene())
        g = ctx.create_map(0, 4096, 1); c = ctx.create_map(1, 4096, 1)
        rp = ctx.params(320, 200, 1)
        good, _ = ctx.render(g, c, rp)
        ctx.set_option("sp_cap_limit", 1500)
        with pytest.raises(ra.PMError, match="shade point queue overflowed"):
            ctx.render(g, c, rp)
        ctx.set_option("sp_cap_limit", 0)
        again, _ = ctx.render(g, c, rp)
        assert np.array_equal(good, again)
        g.destroy(); c.destroy()
    finally:
        ctx.close()


def test_a_failed_map_fails_every_frame_that_uses_it(gpu_ctx):
    """One good map and one assembled from shards that hold too few photons: pm_render reports the bad map on every call
    (not only the first), nothing is left running behind the error, and the good map still renders afterwards."""
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    hip = ctypes.CDLL("libamdhip64.so")
    n, seed = 20000, 5
    cap, per, world = 4096, 2000, 2
    stride = 32 + 32 * cap
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), stride * world) == 0
    for r in range(world):
        gpu_ctx.emit_shard(1, seed, (0.6, 0.6, 0.6), r * per, per, d.value + r * stride, cap)
    bad = gpu_ctx.build_map_shards(d.value, world, stride, 1, n)
    good = gpu_ctx.create_map(0, n, seed, wait=False)
    rp = gpu_ctx.params(320, 200, seed)
    for _ in range(3):
        with pytest.raises(ra.PMError, match="fewer than n photons"):
            gpu_ctx.render(good, bad, rp)
    with pytest.raises(ra.PMError, match="fewer than n photons"):
        bad.download()
    bad.destroy()
    ok = gpu_ctx.create_map(1, n, seed)
    pane, _ = gpu_ctx.render(good, ok, rp)
    again, _ = gpu_ctx.render(good, ok, rp)
    assert np.array_equal(pane, again) and (pane != 0xFF000000).sum() > 1000
    good.destroy(); ok.destroy(); hip.hipFree(d)


def test_narrow_panes(gpu_ctx, orc):
    """Pane widths that are not a multiple of the 8-pixel tile, camera inside the box (every edge tile has shade points)."""
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    n, seed = 3000, 21
    g = gpu_ctx.create_map(0, n, seed); c = gpu_ctx.create_map(1, n, seed)
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    for w, h in ((9, 400), (17, 333), (1, 64), (7, 7)):
        cam = (0.0, 0.0, 1.0)
        pane, _ = gpu_ctx.render(g, c, gpu_ctx.params(w, h, seed, cam=cam))
        want, _ = orc.render(orc.default_scene(), og, oc, orc.params(w, h, seed, cam=cam), nthreads=4)
        assert np.array_equal(pane, want), (w, h)
    og.close(); oc.close(); g.destroy(); c.destroy()


def test_phase_times_of_a_map(gpu_ctx):
    import raytracer_amd as ra
    gpu_ctx.upload_scene(ra.default_scene())
    m = gpu_ctx.create_map(0, 200000, 3, wait=False)
    t_emit, t_build = m.times()
    assert 0 < t_emit < 1000 and 0 < t_build < 1000
    m.destroy()
