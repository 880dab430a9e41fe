"""GPU edge cases through the C-ABI: ragged frame sizes, empty and tiny photon maps, the PM_FLAG_* switches, and the
error channel (the reference has none: every misuse there is undefined behaviour; here it is a status + message)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pane_pair(gpu_ctx, orc, w, h, n, seed, flags=0):
    import raytracer_amd as ra
    g, c = gpu_ctx.create_map(0, n, seed), gpu_ctx.create_map(1, n, seed)
    rp = gpu_ctx.params(w, h, seed, flags=flags)
    pane, st = gpu_ctx.render(None if flags & ra.PM_FLAG_NO_GLOBAL else g, None if flags & ra.PM_FLAG_NO_CAUSTIC else c, rp)
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    want, ost = orc.render(orc.default_scene(), og, oc, orc.params(w, h, seed, flags=flags), nthreads=8)
    g.destroy(); c.destroy(); og.close(); oc.close()
    return pane, want, st, ost


@pytest.mark.parametrize("w,h", [(1, 1), (37, 23), (640, 1), (17, 200), (333, 129)])
def test_ragged_frame_sizes(gpu_ctx, orc, w, h):
    """Widths and heights that fill neither a 16 x 16 pixel block nor a 16-row block."""
    pane, want, st, ost = _pane_pair(gpu_ctx, orc, w, h, 3000, 19)
    assert pane.shape == (h, w) and np.array_equal(pane, want)
    assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)


@pytest.mark.parametrize("n", [0, 1, 62, 63])
def test_empty_and_tiny_maps(gpu_ctx, orc, n):
    """Fewer photons than the estimate asks for (k = 63): find_nearest_photons returns what there is."""
    pane, want, _, _ = _pane_pair(gpu_ctx, orc, 160, 100, n, 3)
    assert np.array_equal(pane, want)
    m = gpu_ctx.create_map(0, n, 3)
    assert len(m) == n and len(m.download()) == n
    found, idx, d2 = gpu_ctx.knn_query(m, [[0.0, -4.9, 5.0], [100.0, 100.0, 100.0]])
    om = orc.Map(m.download())
    f0, _, od2 = om.knn_exact([0.0, -4.9, 5.0])
    assert found[0] == f0 and found[1] == 0          # nothing within max_dist of the far point
    assert np.array_equal(np.sort(d2[0][:f0]), np.sort(np.asarray(od2)[:f0]))
    om.close(); m.destroy()


@pytest.mark.parametrize("flags", [1, 2, 4, 3, 7])
def test_flags(gpu_ctx, orc, flags):
    pane, want, _, _ = _pane_pair(gpu_ctx, orc, 200, 120, 5000, 11, flags=flags)
    assert np.array_equal(pane, want)


def test_row_split_of_a_ragged_frame(gpu_ctx):
    """world = 3 over 5 row blocks (the last one 7 rows): the three ranks' rows make up the single-rank pane."""
    w, h, n, seed = 123, 71, 4000, 5
    g, c = gpu_ctx.create_map(0, n, seed), gpu_ctx.create_map(1, n, seed)
    whole, _ = gpu_ctx.render(g, c, gpu_ctx.params(w, h, seed))
    parts = np.zeros((h, w), dtype=np.uint32)
    for rank in range(3):
        gpu_ctx.render(g, c, gpu_ctx.params(w, h, seed, rank=rank, world=3), pane=parts)
    assert np.array_equal(parts, whole)
    g.destroy(); c.destroy()


def test_errors_are_reported_not_undefined():
    import raytracer_amd as ra
    ctx, other = ra.Context(0), ra.Context(0)
    try:
        with pytest.raises(ra.PMError, match="no scene"):
            ctx.create_map(0, 100, 1)
        ctx.upload_scene(ra.default_scene()); other.upload_scene(ra.default_scene())
        with pytest.raises(ra.PMError, match="kind"):
            ctx.create_map(7, 100, 1)
        bad = ra.default_scene(); bad[3]["kind"] = 9
        with pytest.raises(ra.PMError, match="unknown kind"):
            ctx.upload_scene(bad)
        g, c = ctx.create_map(0, 100, 1), ctx.create_map(1, 100, 1)
        with pytest.raises(ra.PMError, match="another context"):
            other.render(g, c, other.params(32, 32, 1))
        with pytest.raises(ra.PMError, match="missing"):
            ctx.render(None, c, ctx.params(32, 32, 1))
        rp = ctx.params(0, 32, 1)
        with pytest.raises(ra.PMError):
            ctx.render(g, c, rp)
        with pytest.raises(ra.PMError):
            ra.Context(99)
        # the context still works after the errors
        pane, _ = ctx.render(g, c, ctx.params(32, 32, 1))
        assert pane.shape == (32, 32)
        g.destroy(); c.destroy()
    finally:
        ctx.close(); other.close()


@pytest.mark.parametrize("cam,lc", [((1.5, 0.75, -10.0), (0.6, 0.6, 0.6)), ((-2.0, -1.0, -14.0), (0.9, 0.5, 0.2)), ((0.0, 3.0, -6.0), (0.25, 0.25, 1.0))])
def test_other_cameras_and_light_colours(gpu_ctx, orc, cam, lc):
    """renderScene's camera and light colour are arguments here (src/Image.cpp:33, src/Raytrace.cpp:311): other values than
    the reference's own, against the oracle."""
    w, h, n, seed = 240, 150, 8000, 21
    g, c = gpu_ctx.create_map(0, n, seed, lc), gpu_ctx.create_map(1, n, seed, lc)
    for kind, m in ((0, g), (1, c)):
        want, _ = orc.emit(orc.default_scene(), kind, seed, n, light_color=lc)
        assert np.array_equal(m.download().view(np.uint8), np.asarray(want).view(np.uint8))
    pane, st = gpu_ctx.render(g, c, gpu_ctx.params(w, h, seed, cam=cam, light_color=lc))
    og, oc = orc.Map(g.download()), orc.Map(c.download())
    want, ost = orc.render(orc.default_scene(), og, oc, orc.params(w, h, seed, cam=cam, light_color=lc), nthreads=8)
    assert np.array_equal(pane, want)
    assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
    g.destroy(); c.destroy(); og.close(); oc.close()


def test_plane_filter_verdicts_on_adversarial_operands(gpu_ctx, orc):
    """The radiance sum settles the reference's plane filter |k s|^2 > 0.1 (src/Raytrace.cpp:108-113) without its division
    when the verdict is clear and by the reference's own arithmetic otherwise (pm_knn.h).  Here the operands are made
    awkward on purpose: s_z and s . n exactly zero, tiny and huge ratios, normals from 1e-200 to 1 in length, and normals
    scaled so that |k s|^2 lands within 2^-33 .. 2^-27 of 0.1 on either side, or on it.  Bits must equal the oracle's."""
    import raytracer_amd as ra
    rng = np.random.default_rng(77)
    n_ph = 6000
    ph = np.zeros(n_ph, dtype=ra.photon_dtype)
    pos = rng.uniform(-1.0, 1.0, (n_ph, 3)).astype(np.float32)
    pos[::5, 2] = 0.0                                   # a plane of photons at z = 0 exactly
    pos[1::7, 1] = 0.25                                 # and one at y = 0.25
    ph["x"], ph["y"], ph["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
    d = rng.normal(size=(n_ph, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ph["dx"], ph["dy"], ph["dz"] = d[:, 0].astype(np.float32), d[:, 1].astype(np.float32), d[:, 2].astype(np.float32)
    ph["power"] = rng.integers(0, 256, (n_ph, 4), dtype=np.uint8)
    for kind in (0, 1):
        gm, om = gpu_ctx.build_map(ph, kind), orc.Map(ph)
        pts, nrm = [], []
        for q in range(600):
            p = rng.uniform(-0.9, 0.9, 3)
            mode = q % 6
            if mode == 0: p[2] = 0.0                                         # s_z == 0 against the z = 0 photons
            if mode == 1: p[2] = 10.0 ** -rng.integers(20, 300)              # s_z tiny: den underflows or is denormal
            if mode == 2: p[1] = 0.25                                        # s . n == 0 with n = (0, 1, 0) below
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            if mode == 2: n = np.array([0.0, 1.0, 0.0])
            if mode == 3: n *= 10.0 ** -rng.integers(1, 201)                 # num^2 from ordinary to far below 2^-500
            if mode >= 4:
                # aim at the threshold: pick one of the query's own neighbours and scale n so that, for it, the filter's
                # value is 0.1 (1 + delta)
                f, idx, d2 = om.knn_exact(p)
                j = idx[rng.integers(0, max(f - 1, 1))]
                s = p - np.array([ph["x"][j], ph["y"][j], ph["z"][j]], dtype=np.float64)
                den = 3.0 * s[2] * s[2]
                t1 = (np.dot(s, n) / den) ** 2 * np.dot(s, s) if den > 0 else 0.0
                if t1 > 0.1:
                    delta = rng.choice([0.0, 2.0 ** -33, -2.0 ** -33, 2.0 ** -31, -2.0 ** -31, 2.0 ** -29, -2.0 ** -29, 2.0 ** -27, -2.0 ** -27, 1e-15, -1e-15])
                    n *= np.sqrt(0.1 * (1.0 + delta) / t1)
            pts.append(p); nrm.append(n)
        pts, nrm = np.array(pts), np.array(nrm)
        rad = gpu_ctx.knn_gather(gm, pts, nrm)
        want = om.gather(kind, pts, nrm, order=orc.ORDER_CANONICAL)
        assert np.array_equal(rad.view(np.uint64), want.view(np.uint64))
        assert np.isfinite(want).all() and (want > 0).any()
        gm.destroy(); om.close()
