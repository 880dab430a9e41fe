"""CPU tests: the oracle (oracle/pm_oracle.c) against the golden vectors that the COMPILED REFERENCE
produced (oracle/make_golden.sh -> oracle/_ref/ref_harness, see tests/golden/REPORT_*.txt).  These pin
the checker itself; they need neither the reference nor a GPU."""
import math

import numpy as np
import pytest

from .conftest import golden

CONFIGS = [(160, 90, 2048, 12345), (96, 96, 4096, 2024)]


def tag(w, h, n):
    return f"_{w}x{h}_n{n}"


def test_scene_flattening(orc):
    assert orc.default_scene().tobytes() == golden("scene_default.bin", orc.prim_dtype).tobytes()


def test_intersection_kats(orc):
    """SphereObject / PlaneObject::intersect outputs of the reference, incl. source inside a sphere,
    d.a == 0 (inf / NaN paths), mul == 0 self hits and a moving sphere (dt)."""
    kat = golden("kat_intersect.bin", np.float64).reshape(-1, 15)
    prims = list(orc.default_scene()) + [golden("kat_moving_sphere.bin", orc.prim_dtype)[0]]
    hits = 0
    for rec in kat:
        hit, res, nrm = orc.prim_intersect(prims[int(rec[0])], rec[1:4], rec[4:7], rec[7])
        assert hit == bool(rec[8])
        if hit:
            hits += 1
            assert np.array_equal(res.view(np.uint64), rec[9:12].view(np.uint64))
            assert np.array_equal(nrm.view(np.uint64), rec[12:15].view(np.uint64))
    assert 100 < hits < len(kat)


def test_fastfloor(orc):
    L = orc.lib()
    for v, want in [(0.0, 0), (-0.0, 0), (0.5, 0), (-0.5, -1), (1.0, 1), (-1.0, -1), (254.99999999999997, 254), (255.0, 255),
                    (-3.2, -4), (0.6 * 255, 153), (float("nan"), -2 ** 31), (1e30, -2 ** 31)]:
        assert L.orc_fastfloor(v) == want, v


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_primary_hits(orc, w, h, n, seed):
    ids, pts, nrm = orc.primary_hits(orc.default_scene(), w, h, seed)
    g_ids = golden("primary_ids" + tag(w, h, n) + ".bin", np.int8).reshape(h, w)
    assert np.array_equal(ids, g_ids.astype(np.int32))
    hit = g_ids.ravel() >= 0
    assert np.array_equal(pts[hit].view(np.uint64), golden("primary_points" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)[hit].view(np.uint64))
    assert np.array_equal(nrm[hit].view(np.uint64), golden("primary_normals" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)[hit].view(np.uint64))


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
@pytest.mark.parametrize("kind", [0, 1])
def test_emission_libm_matches_reference(orc, w, h, n, seed, kind):
    got, attempts = orc.emit(orc.default_scene(), kind, seed, n, trig=orc.TRIG_LIBM)
    assert got.tobytes() == golden(f"photons_kind{kind}_n{n}.bin", orc.photon_dtype).tobytes()
    assert attempts == golden(f"photons_kind{kind}_n{n}_meta.bin", np.uint64)[0]


def test_det_sincos_close_to_libm(orc):
    rng = np.random.default_rng(1)
    worst = 0.0
    for x in list(rng.uniform(0, 6.2831853, 20000)) + [0.0, 6.2831852999, math.pi / 2, math.pi, 3 * math.pi / 2, 1e-300, 1e-9]:
        s, c = orc.det_sincos(x)
        for got, want in ((s, math.sin(x)), (c, math.cos(x))):
            err = abs(got - want) / max(np.spacing(abs(want)), 1e-300) if abs(want) > 1e-12 else abs(got - want) / 1e-16
            worst = max(worst, err)
    assert worst <= 4.0   # a few ulp of the result (absolute near the zero crossings)


@pytest.mark.parametrize("kind", [0, 1])
def test_emission_det_vs_libm(orc, kind):
    a, _ = orc.emit(orc.default_scene(), kind, 99, 20000, trig=orc.TRIG_LIBM)
    b, _ = orc.emit(orc.default_scene(), kind, 99, 20000, trig=orc.TRIG_DET)
    assert np.array_equal(a["power"], b["power"]) and np.array_equal(a["bounce"], b["bounce"])
    for f in ("x", "y", "z", "dx", "dy", "dz"):
        ulp = np.abs(a[f].view(np.int32).astype(np.int64) - b[f].view(np.int32).astype(np.int64))
        assert ulp.max() <= 2
    differing = sum(int((a[f] != b[f]).sum()) for f in ("x", "y", "z", "dx", "dy", "dz"))
    assert differing < 0.001 * 6 * len(a)


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_knn_heap_arrays_and_radiance(orc, w, h, n, seed):
    """find_nearest_photons on the reference-shaped kd-tree: same heap arrays as the reference; the exact
    k-NN set equals the reference's set; radiance in reference order is bit-identical."""
    pts = golden("knn_points" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    nrm = golden("knn_normals" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    g_idx = golden("knn_ref_idx" + tag(w, h, n) + ".bin", np.int32).reshape(2, -1, 63)
    g_d2 = golden("knn_ref_d2" + tag(w, h, n) + ".bin", np.float64).reshape(2, -1, 63)
    g_found = golden("knn_ref_found" + tag(w, h, n) + ".bin", np.int32).reshape(2, -1)
    g_rad = golden("knn_radiance_reforder" + tag(w, h, n) + ".bin", np.float64).reshape(2, -1, 3)
    for kind in (0, 1):
        ph = golden(f"photons_kind{kind}_n{n}.bin", orc.photon_dtype)
        m = orc.Map(ph, build_ref=True, pivot_seed=seed * 2 + kind)
        for q in range(0, len(pts)):
            f, idx, d2 = m.knn_ref(pts[q])
            assert f == g_found[kind, q]
            assert np.array_equal(idx[:f], g_idx[kind, q, :f]) and np.array_equal(d2[:f].view(np.uint64), g_d2[kind, q, :f].view(np.uint64))
            fe, ie, de = m.knn_exact(pts[q])
            fb, ib, db = m.knn_brute(pts[q])
            assert fe == fb == f and np.array_equal(ie[:fe], ib[:fb])
            assert np.array_equal(np.sort(d2[:f]).view(np.uint64), de[:fe].view(np.uint64))
        rad = m.gather(kind, pts, nrm, order=orc.ORDER_REFERENCE)
        assert np.array_equal(rad.view(np.uint64), g_rad[kind].view(np.uint64))
        can = m.gather(kind, pts, nrm, order=orc.ORDER_CANONICAL)
        assert np.allclose(can, g_rad[kind], rtol=1e-12, atol=1e-15)
        m.close()


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_panes(orc, w, h, n, seed):
    """traceRay under tape replay: the oracle's reference-order pane is the compiled reference's pane;
    the canonical-order pane (what the GPU is held to) and the direct-only pane match their goldens."""
    g = orc.Map(golden(f"photons_kind0_n{n}.bin", orc.photon_dtype), build_ref=True, pivot_seed=seed * 2)
    c = orc.Map(golden(f"photons_kind1_n{n}.bin", orc.photon_dtype), build_ref=True, pivot_seed=seed * 2 + 1)
    scene = orc.default_scene()
    pane, st = orc.render(scene, g, c, orc.params(w, h, seed, order=orc.ORDER_REFERENCE), nthreads=4)
    assert np.array_equal(pane, golden("pane_full_ref" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w))
    assert [st.closest_hit_queries, st.shadow_queries, st.gathers, st.absorb_points] == list(golden("pane_stats" + tag(w, h, n) + ".bin", np.uint64))
    pane, _ = orc.render(scene, g, c, orc.params(w, h, seed, order=orc.ORDER_CANONICAL), nthreads=4)
    assert np.array_equal(pane, golden("pane_full_canonical" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w))
    pane, _ = orc.render(scene, None, None, orc.params(w, h, seed, flags=3), nthreads=4)
    assert np.array_equal(pane, golden("pane_direct" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w))
    # light patch colour (src/Raytrace.cpp:46) appears, background is opaque black
    assert (pane == 0xFFCBCBCB).any() and (pane == 0xFF000000).any()
    g.close(); c.close()


def test_row_job_pool_is_deterministic(orc):
    """One job per row on 1 or 8 workers gives the same pane (the reference's racy global RNG does not)."""
    ph, _ = orc.emit(orc.default_scene(), 0, 5, 3000)
    pc, _ = orc.emit(orc.default_scene(), 1, 5, 3000)
    g, c = orc.Map(ph), orc.Map(pc)
    rp = orc.params(128, 72, 5)
    a, _ = orc.render(orc.default_scene(), g, c, rp, nthreads=1)
    b, _ = orc.render(orc.default_scene(), g, c, rp, nthreads=8)
    assert np.array_equal(a, b)
    g.close(); c.close()


def test_empty_and_tiny_maps(orc):
    m = orc.Map(np.zeros(0, dtype=orc.photon_dtype))
    f, idx, d2 = m.knn_exact([0, 0, 0])
    assert f == 0
    assert np.array_equal(m.gather(0, [[0, 0, 5]], [[0, 1, 0]]), np.zeros((1, 3)))
    m.close()
    ph, _ = orc.emit(orc.default_scene(), 0, 1, 10)
    m = orc.Map(ph)
    f, idx, d2 = m.knn_exact([0, 0, 5])
    assert f == 10 and np.all(np.diff(d2[:10]) >= 0)
    m.close()


def test_byte_over_255_identity():
    """The product's division-free x / 255.0 (pm_knn.h, byte_over_255) and its caustic power factor, for all 256 photon
    power bytes, in exact arithmetic: q = fl(x * inv); res = fl(x - q * 255) (exact in an fma); fl(q + res * inv) must be
    the correctly rounded quotient; min((int)((double)(float(x) / 255.0f) + 0.5), 1) must be x >= 128."""
    from fractions import Fraction
    inv = 1.0 / 255.0
    for x in range(256):
        q = x * inv
        res = float(Fraction(x) - Fraction(q) * 255)
        assert float(Fraction(q) + Fraction(res) * Fraction(inv)) == x / 255.0, x
    xf = np.arange(256, dtype=np.float32)
    factor = np.minimum(((xf / np.float32(255.0)).astype(np.float64) + 0.5).astype(np.int32), 1)
    assert np.array_equal(factor, (np.arange(256) >= 128).astype(np.int32))
