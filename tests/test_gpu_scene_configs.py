"""GPU parity on the synthetic scenes of BASELINE configs 4 and 5 (raytracer_amd/scenes.py), through the C-ABI.

Config 5 (five walls + 10 000 small spheres) is the scene the sphere grid exists for: against the oracle's linear scan
at sizes the oracle finishes in seconds, and at a larger frame against the GPU's own linear scan (PM_SCENE_GRID=0), which
is the same arithmetic without the grid.  Config 4 (five glass spheres) drives the bounce tree to its full depth and
fills the caustic map with photons that passed through glass."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(scene, grid=True):
    import raytracer_amd as ra
    old = os.environ.get("PM_SCENE_GRID")
    os.environ["PM_SCENE_GRID"] = "1" if grid else "0"
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(scene)
    finally:
        if old is None:
            del os.environ["PM_SCENE_GRID"]
        else:
            os.environ["PM_SCENE_GRID"] = old
    return ctx


@pytest.fixture(scope="module")
def c5():
    import raytracer_amd as ra
    scene = ra.many_spheres_scene()
    ctx = _ctx(scene)
    yield ctx, scene
    ctx.close()


def test_c5_primary_hits_bit_exact(c5, orc):
    ctx, scene = c5
    w, h, seed = 256, 160, 5
    ids, pts, nrm = ctx.primary_hits(ctx.params(w, h, seed))
    o_ids, o_pts, o_nrm = orc.primary_hits(scene, w, h, seed)
    assert np.array_equal(ids, o_ids)
    assert len(np.unique(ids)) > 2000                  # thousands of different spheres are seen
    hit = ids.ravel() >= 0
    assert np.array_equal(pts.reshape(-1, 3)[hit].view(np.uint64), o_pts.reshape(-1, 3)[hit].view(np.uint64))
    assert np.array_equal(nrm.reshape(-1, 3)[hit].view(np.uint64), o_nrm.reshape(-1, 3)[hit].view(np.uint64))


@pytest.mark.parametrize("kind", [0, 1])
def test_c5_emission_bit_exact(c5, orc, kind):
    ctx, scene = c5
    n, seed = 3000, 23
    m = ctx.create_map(kind, n, seed)
    want, _ = orc.emit(scene, kind, seed, n)
    assert np.array_equal(m.download().view(np.uint8), np.asarray(want).view(np.uint8))
    m.destroy()


def test_c5_pane_bit_exact(c5, orc):
    ctx, scene = c5
    w, h, n, seed = 96, 64, 3000, 41
    g, c = ctx.create_map(0, n, seed), ctx.create_map(1, n, seed)
    pane, st = ctx.render(g, c, ctx.params(w, h, seed))
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    want, ost = orc.render(scene, og, oc, orc.params(w, h, seed), nthreads=16)
    assert np.array_equal(pane, want)
    assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
    g.destroy(); c.destroy(); og.close(); oc.close()


@pytest.mark.parametrize("loose", ["0", "2", "8"])
def test_c5_spread_out_tiles_through_the_spatial_sort(orc, loose, monkeypatch):
    """Pixel tiles that are deep (several small spheres one behind the other) are not gather groups: their shade points are
    sorted by where they lie and grouped with the bounced pixels' points (PM_LOOSE_TILES: how deep, in tile spans; 0: never;
    unset: decided from the frame before).  Whatever the routing, the pane is the oracle's."""
    import raytracer_amd as ra
    monkeypatch.setenv("PM_LOOSE_TILES", loose)
    scene = ra.many_spheres_scene()
    ctx = _ctx(scene)
    try:
        w, h, n, seed = 192, 128, 20000, 43
        g, c = ctx.create_map(0, n, seed), ctx.create_map(1, n, seed)
        pane, st = ctx.render(g, c, ctx.params(w, h, seed))
        og, oc = orc.Map(g.download()), orc.Map(c.download())
        want, ost = orc.render(scene, og, oc, orc.params(w, h, seed), nthreads=16)
        assert np.array_equal(pane, want)
        assert st.gathers == ost.gathers
        g.destroy(); c.destroy(); og.close(); oc.close()
    finally:
        ctx.close()


def test_c5_frames_in_a_row_agree(c5):
    """The routing of spread-out tiles is decided from the frame before: the first, second and third frame of a scene
    (routing off, on, on) give the same pane."""
    ctx, scene = c5
    w, h, n, seed = 480, 320, 100000, 44
    g, c = ctx.create_map(0, n, seed), ctx.create_map(1, n, seed)
    panes = [ctx.render(g, c, ctx.params(w, h, seed))[0].copy() for _ in range(3)]
    assert np.array_equal(panes[0], panes[1]) and np.array_equal(panes[0], panes[2])
    g.destroy(); c.destroy()


def test_c5_grid_equals_linear_scan_at_size(c5):
    """720 x 720, 200 000 photons per map: hits, photons and pane with the grid = those of the linear scan on the GPU."""
    ctx, scene = c5
    lin = _ctx(scene, grid=False)
    try:
        w, h, n, seed = 720, 720, 200000, 9
        rp = ctx.params(w, h, seed)
        a_ids, a_pts, a_nrm = ctx.primary_hits(rp)
        b_ids, b_pts, b_nrm = lin.primary_hits(rp)
        assert np.array_equal(a_ids, b_ids)
        assert np.array_equal(a_pts.view(np.uint64), b_pts.view(np.uint64)) and np.array_equal(a_nrm.view(np.uint64), b_nrm.view(np.uint64))
        maps = []
        for cx in (ctx, lin):
            maps.append((cx.create_map(0, n, seed, wait=False), cx.create_map(1, n, seed, wait=False)))
        pa, sa = ctx.render(maps[0][0], maps[0][1], rp)
        pb, sb = lin.render(maps[1][0], maps[1][1], rp)
        for k in (0, 1):
            assert np.array_equal(maps[0][k].download().view(np.uint8), maps[1][k].download().view(np.uint8))
        assert np.array_equal(pa, pb)
        assert (sa.closest_hit_queries, sa.shadow_queries, sa.gathers) == (sb.closest_hit_queries, sb.shadow_queries, sb.gathers)
        for pair in maps:
            for m in pair:
                m.destroy()
    finally:
        lin.close()


def test_c4_caustics_scene_bit_exact(orc):
    import raytracer_amd as ra
    scene = ra.caustics_scene()
    ctx = _ctx(scene)
    try:
        w, h, n, seed = 200, 128, 20000, 77
        ids, pts, nrm = ctx.primary_hits(ctx.params(w, h, seed))
        o_ids, o_pts, o_nrm = orc.primary_hits(scene, w, h, seed)
        assert np.array_equal(ids, o_ids)
        maps, omaps = [], []
        for kind in (0, 1):
            m = ctx.create_map(kind, n, seed)
            want, _ = orc.emit(scene, kind, seed, n)
            got = m.download()
            assert np.array_equal(got.view(np.uint8), np.asarray(want).view(np.uint8))
            maps.append(m); omaps.append(orc.Map(got))
        pane, st = ctx.render(maps[0], maps[1], ctx.params(w, h, seed))
        want, ost = orc.render(scene, omaps[0], omaps[1], orc.params(w, h, seed), nthreads=16)
        assert np.array_equal(pane, want)
        assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
        assert st.complex_pixels > 0.03 * w * h          # glass spheres: every pixel that sees one is a bounce tree
        for m in maps:
            m.destroy()
        for m in omaps:
            m.close()
    finally:
        ctx.close()
