"""GPU parity on a synthetic many-sphere scene (the shape of BASELINE config 5 at a size the oracle finishes in
seconds): 400 primitives = 58 KB of scene, beyond the 48 KB the kernels stage in LDS, so every kernel reads the scene
from global memory -- once through the sphere grid (the default for 64 spheres and more) and once by the linear scan
(PM_SCENE_GRID=0).  Large, overlapping, partly moving spheres, mirrors and glass: the grid must never change a hit.
Bars as everywhere: bit-exact hits, photons and panes against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def many_spheres(n_spheres=395, seed=17):
    import raytracer_amd as ra
    rng = np.random.default_rng(seed)
    base = ra.default_scene()
    s = np.zeros(5 + n_spheres, dtype=ra.prim_dtype)
    s[:5] = base[:5]                                   # the five walls
    for i in range(n_spheres):
        p = s[5 + i]
        p["kind"] = 0
        p["x"], p["y"], p["z"] = rng.uniform(-4.3, 4.3), rng.uniform(-4.3, 4.3), rng.uniform(1.0, 9.0)
        p["radius"] = rng.uniform(0.15, 0.45)
        if i % 7 == 0:                                 # some move (Sphere::intersect's dt path)
            p["dx"], p["dy"], p["dz"] = rng.uniform(-0.2, 0.2, 3)
        p["red"], p["green"], p["blue"] = rng.uniform(0.2, 1.0, 3).astype(np.float32)
        style = i % 4
        if style == 0:                                 # mirror
            p["specular_chance"] = 1.0
        elif style == 1:                               # glass
            p["specular_chance"], p["transmission_chance"], p["refraction"] = 0.1, 0.9, 1.5
        else:                                          # diffuse
            p["diffuse_chance"], p["absorb_chance"] = 0.5, 1.0
    return s


@pytest.fixture(scope="module", params=["grid", "linear"])
def ctx_scene(request):
    import raytracer_amd as ra
    scene = many_spheres()
    assert scene.nbytes > 48 * 1024
    old = os.environ.get("PM_SCENE_GRID")
    os.environ["PM_SCENE_GRID"] = "1" if request.param == "grid" else "0"      # read by pm_scene_upload
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(scene)
    finally:
        if old is None:
            del os.environ["PM_SCENE_GRID"]
        else:
            os.environ["PM_SCENE_GRID"] = old
    yield ctx, scene
    ctx.close()


def test_primary_hits_bit_exact(ctx_scene, orc):
    ctx, scene = ctx_scene
    w, h, seed = 320, 200, 5
    ids, pts, nrm = ctx.primary_hits(ctx.params(w, h, seed))
    o_ids, o_pts, o_nrm = orc.primary_hits(scene, w, h, seed)
    assert np.array_equal(ids, o_ids)
    assert len(np.unique(ids)) > 100                   # the spheres are really seen
    hit = ids.ravel() >= 0
    assert np.array_equal(pts.reshape(-1, 3)[hit].view(np.uint64), o_pts.reshape(-1, 3)[hit].view(np.uint64))
    assert np.array_equal(nrm.reshape(-1, 3)[hit].view(np.uint64), o_nrm.reshape(-1, 3)[hit].view(np.uint64))


@pytest.mark.parametrize("kind", [0, 1])
def test_emission_bit_exact(ctx_scene, orc, kind):
    ctx, scene = ctx_scene
    n, seed = 4000, 23
    m = ctx.create_map(kind, n, seed)
    want, _ = orc.emit(scene, kind, seed, n)
    assert np.array_equal(m.download().view(np.uint8), np.asarray(want).view(np.uint8))
    m.destroy()


def test_pane_bit_exact(ctx_scene, orc):
    ctx, scene = ctx_scene
    w, h, n, seed = 160, 100, 6000, 41
    g, c = ctx.create_map(0, n, seed), ctx.create_map(1, n, seed)
    pane, st = ctx.render(g, c, ctx.params(w, h, seed))
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    want, ost = orc.render(scene, og, oc, orc.params(w, h, seed), nthreads=8)
    assert np.array_equal(pane, want)
    assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
    assert st.complex_pixels > 0
    g.destroy(); c.destroy(); og.close(); oc.close()
