"""CPU tests: the oracle against the compiled reference on scenes other than the Cornell box of src/Image.cpp.

oracle/ref_harness.cpp builds four more scenes with the reference's own SphereObject / PlaneObject constructors,
runs Scene::intersect, createPhotonMap / createCausticPhotonMap and traceRay on them under tape replay, checks the
restatement bit for bit in-process (tests/golden/REPORT_<scene>_*.txt) and writes the vectors replayed here:
  caustics    BASELINE config 4: five spheres of the reference's glass
  spheres100  BASELINE config 5's generator at 100 spheres
  mixed100    100 large overlapping spheres, every fourth one glass, every fourth one a mirror
  moving100   the same recipe, two thirds of the spheres in linear motion over the exposure (dt)
The flattened scenes (scene_<name>.bin) are what the reference traced; the Python generators must reproduce them."""
import numpy as np
import pytest

from .conftest import golden

SCENES = ["caustics", "spheres100", "mixed100", "moving100"]
W, H, N, SEED = 96, 54, 2048, 4242


def tag(scene):
    return f"_{scene}_{W}x{H}_n{N}"


def test_python_generators_reproduce_the_reference_built_scenes(orc):
    import raytracer_amd as ra
    assert ra.prim_dtype.itemsize == orc.prim_dtype.itemsize == 144
    assert ra.caustics_scene().tobytes() == golden("scene_caustics.bin", np.uint8).tobytes()
    assert ra.many_spheres_scene(100, 2024).tobytes() == golden("scene_spheres100.bin", np.uint8).tobytes()
    mixed = golden("scene_mixed100.bin", orc.prim_dtype)
    sph = mixed[5:]
    assert len(mixed) == 105 and (sph["kind"] == 0).all() and (mixed[:5]["kind"] == 1).all()
    assert (sph["transmission_chance"][0::4] == 0.9).all() and (sph["refraction"][0::4] == 2.5).all()
    assert (sph["specular_chance"][1::4] == 1.0).all() and (sph["diffuse_chance"][2::4] == 0.4).all()
    assert 0.15 <= sph["radius"].min() and sph["radius"].max() <= 0.6


@pytest.mark.parametrize("scene", SCENES)
def test_primary_hits(orc, scene):
    prims = golden(f"scene_{scene}.bin", orc.prim_dtype)
    ids, pts, nrm = orc.primary_hits(prims, W, H, SEED)
    g_ids = golden("primary_ids" + tag(scene) + ".bin", np.int8).reshape(H, W)
    assert np.array_equal(ids, g_ids.astype(np.int32))
    hit = g_ids.ravel() >= 0
    assert np.array_equal(pts[hit].view(np.uint64), golden("primary_points" + tag(scene) + ".bin", np.float64).reshape(-1, 3)[hit].view(np.uint64))
    assert np.array_equal(nrm[hit].view(np.uint64), golden("primary_normals" + tag(scene) + ".bin", np.float64).reshape(-1, 3)[hit].view(np.uint64))
    assert (g_ids >= 5).any()                      # spheres are seen


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("kind", [0, 1])
def test_emission_libm_matches_reference(orc, scene, kind):
    prims = golden(f"scene_{scene}.bin", orc.prim_dtype)
    got, attempts = orc.emit(prims, kind, SEED, N, trig=orc.TRIG_LIBM)
    assert got.tobytes() == golden(f"photons_kind{kind}_{scene}_n{N}.bin", orc.photon_dtype).tobytes()
    assert attempts == golden(f"photons_kind{kind}_{scene}_n{N}_meta.bin", np.uint64)[0]


@pytest.mark.parametrize("scene", SCENES)
def test_panes(orc, scene):
    """traceRay of the compiled reference on the scene == the oracle's reference-order pane, with the same ray counts."""
    prims = golden(f"scene_{scene}.bin", orc.prim_dtype)
    g = orc.Map(golden(f"photons_kind0_{scene}_n{N}.bin", orc.photon_dtype), build_ref=True, pivot_seed=SEED * 2)
    c = orc.Map(golden(f"photons_kind1_{scene}_n{N}.bin", orc.photon_dtype), build_ref=True, pivot_seed=SEED * 2 + 1)
    pane, st = orc.render(prims, g, c, orc.params(W, H, SEED, order=orc.ORDER_REFERENCE), nthreads=4)
    assert np.array_equal(pane, golden("pane_full_ref" + tag(scene) + ".bin", np.uint32).reshape(H, W))
    assert [st.closest_hit_queries, st.shadow_queries, st.gathers, st.absorb_points] == list(golden("pane_stats" + tag(scene) + ".bin", np.uint64))
    pane, _ = orc.render(prims, g, c, orc.params(W, H, SEED, order=orc.ORDER_CANONICAL), nthreads=4)
    assert np.array_equal(pane, golden("pane_full_canonical" + tag(scene) + ".bin", np.uint32).reshape(H, W))
    pane, _ = orc.render(prims, None, None, orc.params(W, H, SEED, flags=3), nthreads=4)
    assert np.array_equal(pane, golden("pane_direct" + tag(scene) + ".bin", np.uint32).reshape(H, W))
    g.close(); c.close()
