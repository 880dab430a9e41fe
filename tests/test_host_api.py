"""The reference's host API above the C-ABI (raytracer_amd/host/raytrace_api.h): a caller written in the shape of the
reference's own call sites (tests/host/host_caller.cpp: Vec3::normalize, scene->intersect, obj->intersect,
createPhotonMap, find_nearest_photons, insert, the scene constructors and the scene loader) is compiled against
libraytrace_host.so here and must produce the oracle's bits.  CPU tests: what needs no GPU (scene records, the heap);
GPU tests: the intersect calls and the photon-map calls."""
import os
import subprocess

import numpy as np
import pytest

from .conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "raytracer_amd")
EXE = os.path.join(ROOT, "raytracer_amd", "host", "raytracer")


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    """tests/host/host_caller.cpp built against the host API library (g++, seconds)."""
    if not os.path.exists(os.path.join(LIBDIR, "libraytrace_host.so")):
        pytest.skip("libraytrace_host.so not built (make -C raytracer_amd/host)")
    exe = str(tmp_path_factory.mktemp("host") / "host_caller")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "host", "host_caller.cpp"),
                    "-L" + LIBDIR, "-lraytrace_host", "-lraytracer_amd", "-Wl,-rpath," + LIBDIR], check=True)
    return exe


def call(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    return r


def test_scene_constructors_and_generators(caller, tmp_path):
    """Scenes built with the reference's constructors in C++ flatten to the records the Python side uploads."""
    import raytracer_amd as ra
    for spec, want in (("default", ra.default_scene()), ("caustics", ra.caustics_scene()),
                       ("spheres:10000:2024", ra.many_spheres_scene()), ("spheres:37:99", ra.many_spheres_scene(37, 99))):
        out = tmp_path / "scene.bin"
        call(caller, "dump", spec, out)
        assert np.fromfile(out, dtype=ra.prim_dtype).tobytes() == want.tobytes(), spec


def test_scene_files(caller, tmp_path):
    """loadScene: binary files are pm_prim records as they are; a text file goes through the reference's constructors."""
    import raytracer_amd as ra
    for name in ("default", "caustics", "mixed100", "moving100"):
        out = tmp_path / "scene.bin"
        call(caller, "dump", os.path.join(ROOT, "tests", "golden", f"scene_{name}.bin"), out)
        assert np.fromfile(out, dtype=ra.prim_dtype).tobytes() == golden(f"scene_{name}.bin", ra.prim_dtype).tobytes(), name
    # the reference's Cornell box (src/Image.cpp:20-28) and its commented-out moving sphere (:29-30), as text
    text = tmp_path / "cornell.txt"
    text.write_text("""# x y z  min max  colour  d s t a
plane 0 -5 0  -5 5  FFEEEEEE  0.4 0 0 1
plane 0 5 0   -5 5  FDEEEEEE  0.4 0 0 1     # the 9-digit literal of the reference, truncated
plane 5 0 0   -5 5  FFFF3333  0.4 0 0 1
plane -5 0 0  -5 5  FF3333FF  0.4 0 0 1
plane 0 0 10  -5 5  FFEEEEEE  0.4 0 0 1

sphere 2 -3.5 3   1.5  FFFFFFFF  0.0 0.1 0.9 0.0  2.5
sphere -2 -3.5 5  1.5  FFFFFFFF  0.0 1.0 0.0 0.0
sphere 0 1 5      0.8  FF33FF33  0.4 0.0 0.0 1.0  0  0.2 0 0
""")
    out = tmp_path / "scene.bin"
    call(caller, "dump", text, out)
    got = np.fromfile(out, dtype=ra.prim_dtype)
    assert got[:7].tobytes() == ra.default_scene().tobytes()
    m = got[7]
    assert (m["kind"], m["x"], m["y"], m["z"], m["radius"], m["dx"], m["dy"], m["dz"]) == (0, 0.0, 1.0, 5.0, 0.8, 0.2, 0.0, 0.0)
    assert m["green"] == np.float32(255) / np.float32(255.0) and m["red"] == np.float32(0x33) / np.float32(255.0) and m["absorb_chance"] == 1.0
    bad = tmp_path / "bad.txt"
    bad.write_text("sphere 1 2\n")
    r = subprocess.run([caller, "dump", str(bad), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "bad.txt:1" in r.stderr


def test_heap_insert_matches_the_oracle(caller, tmp_path, orc):
    """raytrace::insert (src/PhotonMap.cpp:142-206) in the call pattern of find_nearest_photons, many ties: the heap arrays
    equal those of the oracle's restatement (which the reference's own heap arrays pin, tests/golden/knn_ref_*)."""
    import ctypes as C
    L = orc.lib()
    for n, seed in ((40, 1), (63, 2), (64, 3), (5000, 4), (20000, 5)):
        out = tmp_path / "heap.bin"
        call(caller, "heap", n, seed, out)
        got = np.fromfile(out, dtype=[("who", "<i8"), ("d", "<f8")])
        nearest = np.full(63, -1, dtype=np.int32); dist = np.zeros(63)
        state, size, mask = seed, 0, (1 << 64) - 1
        for i in range(n):
            state = (state + 0x9E3779B97F4A7C15) & mask
            z = state
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
            z ^= z >> 31
            d = (z % 997) / 8.0
            if size < 63:
                L.orc_heap_insert(nearest.ctypes.data, dist.ctypes.data, 63, size, i, d); size += 1
            elif d < dist[0]:
                L.orc_heap_insert(nearest.ctypes.data, dist.ctypes.data, 63, size, i, d)
        assert np.array_equal(got["who"], nearest.astype(np.int64)) and np.array_equal(got["d"].view(np.uint64), dist.view(np.uint64)), (n, seed)
        # and the property the domain offers: a max-heap holding the min(n, 63) smallest distances
        k = min(n, 63)
        assert all(dist[(i - 1) // 2] >= dist[i] for i in range(1, k))


@pytest.mark.gpu
def test_scene_intersect_through_the_host_api(caller, tmp_path, orc):
    """scene->intersect / obj->intersect with rays made by Vec3 arithmetic on the host: ids, points and normals of the
    primary hit and of its mirror bounce (exclude = the object hit) equal the oracle's Scene::intersect, bit for bit."""
    w, h = 64, 36
    out = tmp_path / "trace.bin"
    call(caller, "trace", w, h, out)
    rec = np.fromfile(out, dtype=np.float64).reshape(h, w, 14)
    scene = orc.default_scene()
    cam = np.array([0.0, 0.0, -12.0])
    fov = (w / 1280.0) * 64.0
    hits = bounced = 0
    for y in range(h):
        for x in range(w):
            ray = np.array([(x - w // 2) / fov - cam[0], (y - h // 2) / fov - cam[1], -cam[2]])
            ray = ray / np.sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2])
            obj, res, nrm = orc.scene_intersect(scene, cam, ray, -1, 0.25)
            r = rec[y, x]
            assert int(r[0]) == obj
            if obj < 0:
                continue
            hits += 1
            assert np.array_equal(r[1:4].view(np.uint64), res.view(np.uint64)) and np.array_equal(r[4:7].view(np.uint64), nrm.view(np.uint64))
            k = nrm[0] * ray[0] + nrm[1] * ray[1] + nrm[2] * ray[2]
            n1 = nrm * k
            n1 = n1 * 2
            ray2 = ray - n1
            ray2 = ray2 / np.sqrt(ray2[0] * ray2[0] + ray2[1] * ray2[1] + ray2[2] * ray2[2])
            obj2, res2, nrm2 = orc.scene_intersect(scene, res, ray2, obj, 0.25)
            assert int(r[7]) == obj2          # -99 would mean obj->intersect disagreed with scene->intersect
            if obj2 >= 0:
                bounced += 1
                assert np.array_equal(r[8:11].view(np.uint64), res2.view(np.uint64)) and np.array_equal(r[11:14].view(np.uint64), nrm2.view(np.uint64))
    assert hits > 200 and bounced > 100


@pytest.mark.gpu
def test_photon_map_calls_through_the_host_api(caller, tmp_path, orc):
    """createPhotonMap + find_nearest_photons as the reference's own code calls them: found, the squared distances and the
    photon records behind the returned pointers equal the oracle's exact 63-NN on the oracle's photons."""
    import raytracer_amd as ra
    n, seed, m = 5000, 31337, 40
    out = tmp_path / "knn.bin"
    r = call(caller, "knn", n, seed, m, out)
    assert "Global photons traced in" in r.stdout and "Global photons kd-tree built in" in r.stdout
    rec = np.fromfile(out, dtype=np.dtype([("q", "<f8", 3), ("found", "<f8"), ("nn", [("d", "<f8"), ("ph", ra.photon_dtype)], 63)]))
    assert len(rec) == m
    photons, _ = orc.emit(orc.default_scene(), 0, seed, n)
    omap = orc.Map(photons)
    for row in rec:
        found, idx, d2 = omap.knn_exact(row["q"])
        assert int(row["found"]) == found == 63
        assert np.array_equal(row["nn"]["d"].view(np.uint64), d2.view(np.uint64))
        assert row["nn"]["ph"].tobytes() == photons[idx].tobytes()
    omap.close()


@pytest.mark.gpu
def test_find_nearest_photons_with_a_smaller_max_dist(caller, tmp_path, orc):
    """max_dist below the reference's 100 (its `dist < max_dist`, src/PhotonMap.cpp:215): the photons closer than that among the
    exact 63 nearest, in ascending order, and nothing else."""
    import raytracer_amd as ra
    n, seed, m, max_dist = 5000, 31337, 40, 0.05
    out = tmp_path / "knn_md.bin"
    call(caller, "knn", n, seed, m, out, max_dist)
    rec = np.fromfile(out, dtype=np.dtype([("q", "<f8", 3), ("found", "<f8"), ("nn", [("d", "<f8"), ("ph", ra.photon_dtype)], 63)]))
    photons, _ = orc.emit(orc.default_scene(), 0, seed, n)
    omap = orc.Map(photons)
    cut = 0
    for row in rec:
        found, idx, d2 = omap.knn_exact(row["q"])
        want = int(np.count_nonzero(d2[:found] < max_dist))
        assert int(row["found"]) == want
        assert np.array_equal(row["nn"]["d"][:want].view(np.uint64), d2[:want].view(np.uint64))
        assert row["nn"]["ph"][:want].tobytes() == photons[idx[:want]].tobytes()
        cut += want < found
    omap.close()
    assert 0 < cut        # the test means something: some searches were cut short


@pytest.mark.gpu
@pytest.mark.parametrize("scene,samples", [("mixed100", 1), ("caustics", 3)])
def test_cli_renders_a_scene_file(tmp_path, orc, scene, samples):
    """raytracer --scene FILE: the PNG equals the oracle's pane of that scene through the front-end (box filter, flip, byte order)."""
    from PIL import Image
    import raytracer_amd as ra
    w, h, n, seed = 160, 90, 2048, 99
    e = dict(os.environ, RAYTRACER_SEED=str(seed), RAYTRACER_PHOTONS=str(n))
    r = subprocess.run(
                       [EXE, "--scene", os.path.join(ROOT, "tests", "golden", f"scene_{scene}.bin"), "8", str(w), str(h), str(samples)],
                       cwd=tmp_path, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = np.array(Image.open(tmp_path / "raytraced.png"))
    prims = golden(f"scene_{scene}.bin", orc.prim_dtype)
    g, c = orc.Map(orc.emit(prims, 0, seed, n)[0]), orc.Map(orc.emit(prims, 1, seed, n)[0])
    pane, _ = orc.render(prims, g, c, orc.params(w * samples, h * samples, seed), nthreads=8)
    ch = [((pane >> sh) & 0xFF).astype(np.int64).reshape(h, samples, w, samples).sum(axis=(1, 3)) // (samples * samples) for sh in (16, 8, 0)]
    red, green, blue = [x[::-1] for x in ch]
    assert np.array_equal(img[..., 0], blue) and np.array_equal(img[..., 1], green) and np.array_equal(img[..., 2], red)
    g.close(); c.close()


def _run_ranks(tmp_path, world, args, env, out_name):
    """One raytracer process per rank, all started at once (one process per GPU: rank r sees GPU r as device 0)."""
    id_file = tmp_path / f"rccl_id_{out_name}"
    id_file.write_bytes(b"\x01" * 136)       # what a rank 0 that died would have left: token 0x0101..., not ours
    procs = []
    for r in range(world):
        e = dict(env, HIP_VISIBLE_DEVICES=str(r)) if world > 1 else dict(env)
        # (the token tells this launch's id file from one a dead rank 0 left behind; a stale file with another token is waiting there)
        procs.append(subprocess.Popen([EXE, "--rank", str(r), "--world", str(world), "--id-file", str(id_file), "--token", str(os.getpid()), "--out", out_name] + args,
                                      cwd=tmp_path, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_cli_one_process_per_gpu_over_rccl(tmp_path, world):
    """The C++ multi-GPU frame (raytrace_dist.h: attempt ranges per rank, one RCCL all-gather of the photon shards per map,
    interleaved row blocks sent to rank 0): the PNG of N ranks equals the PNG of the single-GPU path byte for byte.
    One rank exercises RCCL and the shard path on any box; two need two GPUs."""
    import torch
    if world > torch.cuda.device_count():
        pytest.skip(f"{world} GPUs needed, {torch.cuda.device_count()} visible")
    w, h, s, n, seed = 320, 180, 2, 20000, 31
    e = dict(os.environ, RAYTRACER_SEED=str(seed), RAYTRACER_PHOTONS=str(n))
    args = ["--scene", "caustics", "8", str(w), str(h), str(s)]
    r = subprocess.run([EXE, "--out", "single.png"] + args, cwd=tmp_path, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    outs = _run_ranks(tmp_path, world, args, e, "ranks.png")
    assert "photon maps" in outs[0][0] and f"on {world} ranks" in outs[0][0]
    assert (tmp_path / "ranks.png").read_bytes() == (tmp_path / "single.png").read_bytes()
