"""The synthetic scenes of BASELINE configs 4 and 5 (raytracer_amd/scenes.py): shape, determinism, and that the oracle
traces them (CPU only)."""
import numpy as np


def test_many_spheres_scene_is_deterministic_and_in_range():
    import raytracer_amd as ra
    a, b = ra.many_spheres_scene(), ra.many_spheres_scene()
    assert a.tobytes() == b.tobytes() and len(a) == 10005
    assert a.tobytes() != ra.many_spheres_scene(seed=1).tobytes()
    sp = a[5:]
    assert (a[:5]["kind"] == 1).all() and (sp["kind"] == 0).all()
    for ax, lo, hi in (("x", -4.5, 4.5), ("y", -4.5, 4.5), ("z", 0.5, 9.5), ("radius", 0.02, 0.08)):
        assert sp[ax].min() >= lo and sp[ax].max() <= hi and sp[ax].max() - sp[ax].min() > 0.95 * (hi - lo)
    assert (sp["absorb_chance"] == 1.0).all() and (sp["diffuse_chance"] == 0.4).all()
    byte = sp["red"].astype(np.float64) * 255
    assert np.allclose(byte, np.round(byte), atol=1e-4) and byte.min() >= 64 and byte.max() <= 255


def test_uniform_stream_known_values():
    from raytracer_amd.scenes import uniform_stream
    u = uniform_stream(2024, 4)
    # pure-Python restatement of the stream definition
    M = (1 << 64) - 1

    def mix(z):
        z ^= z >> 30; z = z * 0xBF58476D1CE4E5B9 & M
        z ^= z >> 27; z = z * 0x94D049BB133111EB & M
        return z ^ (z >> 31)
    base = mix(2024)
    want = [(mix((base + (j + 1) * 0x9E3779B97F4A7C15) & M) >> 11) / 9007199254740992.0 for j in range(4)]
    assert u.tolist() == want


def test_caustics_scene_spheres_are_disjoint_and_inside_the_box():
    import raytracer_amd as ra
    from raytracer_amd.scenes import GLASS_SPHERES
    s = ra.caustics_scene()
    assert len(s) == 10 and (s[5:]["transmission_chance"] == 0.9).all() and (s[5:]["refraction"] == 2.5).all()
    for i, (x, y, z, r) in enumerate(GLASS_SPHERES):
        assert abs(x) + r <= 5 and abs(y) + r <= 5 and 0 <= z - r and z + r <= 10
        for (x2, y2, z2, r2) in GLASS_SPHERES[i + 1:]:
            assert np.sqrt((x - x2) ** 2 + (y - y2) ** 2 + (z - z2) ** 2) > r + r2


def test_oracle_traces_the_synthetic_scenes(orc):
    import raytracer_amd as ra
    for scene, lo in ((ra.many_spheres_scene(500), 15), (ra.caustics_scene(), 5)):
        ids, _, _ = orc.primary_hits(scene, 64, 40, 3)
        assert len(np.unique(ids)) > lo
