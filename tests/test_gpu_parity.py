"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle and the golden
vectors produced by the compiled reference (tests/golden/, see oracle/make_golden.sh).

Bars: bit-exact for hit ids, hit points / normals, photon records, k-NN sets and distances, radiance
estimates (canonical summation order) and panes; PSNR >= 40 dB against the reference-order panes.
"""
import numpy as np
import pytest

from .conftest import golden, psnr

pytestmark = pytest.mark.gpu

CONFIGS = [(160, 90, 2048, 12345), (96, 96, 4096, 2024)]


def tag(w, h, n):
    return f"_{w}x{h}_n{n}"


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_primary_hits_match_reference_golden(gpu_ctx, w, h, n, seed):
    rp = gpu_ctx.params(w, h, seed)
    ids, pts, nrm = gpu_ctx.primary_hits(rp)
    g_ids = golden("primary_ids" + tag(w, h, n) + ".bin", np.int8).reshape(h, w)
    g_pts = golden("primary_points" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    g_nrm = golden("primary_normals" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    assert np.array_equal(ids, g_ids.astype(np.int32))
    hit = (g_ids.ravel() >= 0)
    assert np.array_equal(pts[hit].view(np.uint64), g_pts[hit].view(np.uint64))
    assert np.array_equal(nrm[hit].view(np.uint64), g_nrm[hit].view(np.uint64))


def test_primary_hits_1440_bit_exact(gpu_ctx, orc):
    """BASELINE config 2 size: every primary hit id, point and normal equals the oracle's."""
    w = h = 1440
    rp = gpu_ctx.params(w, h, 12345)
    ids, pts, nrm = gpu_ctx.primary_hits(rp)
    o_ids, o_pts, o_nrm = orc.primary_hits(orc.default_scene(), w, h, 12345)
    assert np.array_equal(ids, o_ids)
    hit = o_ids.ravel() >= 0
    assert np.array_equal(pts[hit].view(np.uint64), o_pts[hit].view(np.uint64))
    assert np.array_equal(nrm[hit].view(np.uint64), o_nrm[hit].view(np.uint64))
    hist = np.bincount(ids.ravel() + 1, minlength=8)
    assert hist[0] > 0.7 * w * h  # ~75 % of primary rays miss at 1440x1440


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 2048, 50000])
def test_emission_bit_exact_vs_oracle(gpu_ctx, orc, kind, n):
    m = gpu_ctx.create_map(kind, n, 777)
    got = m.download()
    m.destroy()
    want, _ = orc.emit(orc.default_scene(), kind, 777, n, trig=orc.TRIG_DET)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("kind", [0, 1])
def test_emission_vs_reference_golden(gpu_ctx, kind):
    """The reference calls libm sin/cos; the product uses a deterministic sin/cos.  Records may differ
    in the last float ulp of a coordinate, nothing else."""
    g = golden(f"photons_kind{kind}_n2048.bin", __import__("raytracer_amd").photon_dtype)
    m = gpu_ctx.create_map(kind, 2048, 12345)
    got = m.download()
    m.destroy()
    assert np.array_equal(got["power"], g["power"]) and np.array_equal(got["bounce"], g["bounce"])
    for f in ("x", "y", "z", "dx", "dy", "dz"):
        ulp = np.abs(got[f].view(np.int32).astype(np.int64) - g[f].view(np.int32).astype(np.int64))
        assert ulp.max() <= 2, (f, ulp.max())
    # and the difference is rare: all but a handful of the 6 x 2048 coordinates are the reference's very bits
    differing = sum(int((got[f] != g[f]).sum()) for f in ("x", "y", "z", "dx", "dy", "dz"))
    assert differing <= 6 * 2048 // 100, differing


def _query_points(orc, w, h, seed, step=5):
    ids, pts, nrm = orc.primary_hits(orc.default_scene(), w, h, seed)
    sel = np.flatnonzero(ids.ravel() >= 0)[::step]
    extra = np.array([[0, 0, 5], [4.9, 4.9, 9.9], [-4.99, -4.99, 0.01], [0, 0, -12], [30, 30, 30]], dtype=np.float64)
    en = np.tile(np.array([[0.0, 1.0, 0.0]]), (len(extra), 1))
    return np.concatenate([pts[sel], extra]), np.concatenate([nrm[sel], en])


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("n", [10, 63, 64, 100, 2048, 20000, 300000])
def test_knn_and_gather_bit_exact_vs_oracle(gpu_ctx, orc, kind, n):
    photons, _ = orc.emit(orc.default_scene(), kind, 4242, n, trig=orc.TRIG_DET)
    gm = gpu_ctx.build_map(photons, kind)
    om = orc.Map(photons)
    pts, nrm = _query_points(orc, 96, 96, 5, step=9)
    found, idx, d2 = gpu_ctx.knn_query(gm, pts)
    for q in range(len(pts)):
        f, oi, od = om.knn_exact(pts[q])
        assert found[q] == f
        assert np.array_equal(d2[q, :f].view(np.uint64), od[:f].view(np.uint64))
        if not np.array_equal(idx[q, :f], oi[:f]):
            # only photons at exactly the same distance may swap, and only the k-th slot may differ in identity
            assert np.array_equal(idx[q, :f - 1], oi[:f - 1])
    rad = gpu_ctx.knn_gather(gm, pts, nrm)
    want = om.gather(kind, pts, nrm, order=orc.ORDER_CANONICAL)
    assert np.array_equal(rad.view(np.uint64), want.view(np.uint64))
    gm.destroy(); om.close()


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_gather_vs_reference_golden(gpu_ctx, w, h, n, seed):
    """Radiance estimates on the reference's own photons and heap order (golden) vs the GPU's canonical
    order: same k-NN distances, sums equal up to floating-point summation order."""
    import raytracer_amd as ra
    pts = golden("knn_points" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    nrm = golden("knn_normals" + tag(w, h, n) + ".bin", np.float64).reshape(-1, 3)
    g_d2 = golden("knn_ref_d2" + tag(w, h, n) + ".bin", np.float64).reshape(2, -1, 63)
    g_found = golden("knn_ref_found" + tag(w, h, n) + ".bin", np.int32).reshape(2, -1)
    g_rad = golden("knn_radiance_reforder" + tag(w, h, n) + ".bin", np.float64).reshape(2, -1, 3)
    for kind in (0, 1):
        photons = golden(f"photons_kind{kind}_n{n}.bin", ra.photon_dtype)
        gm = gpu_ctx.build_map(photons, kind)
        found, idx, d2 = gpu_ctx.knn_query(gm, pts)
        assert np.array_equal(found, g_found[kind])
        for q in range(len(pts)):
            f = found[q]
            assert np.array_equal(np.sort(g_d2[kind, q, :f]).view(np.uint64), d2[q, :f].view(np.uint64))
        rad = gpu_ctx.knn_gather(gm, pts, nrm)
        assert np.allclose(rad, g_rad[kind], rtol=1e-12, atol=1e-15)
        gm.destroy()


@pytest.mark.parametrize("w,h,n,seed", CONFIGS)
def test_panes_vs_reference_golden(gpu_ctx, w, h, n, seed):
    """Direct-lighting pass and full pane on the reference's photons: bit-exact against the oracle's
    canonical-order pane, >= 40 dB (in fact identical or 1 LSB) against the compiled reference's pane."""
    import raytracer_amd as ra
    gm = gpu_ctx.build_map(golden(f"photons_kind0_n{n}.bin", ra.photon_dtype), 0)
    cm = gpu_ctx.build_map(golden(f"photons_kind1_n{n}.bin", ra.photon_dtype), 1)
    rp = gpu_ctx.params(w, h, seed, flags=ra.PM_FLAG_NO_GLOBAL | ra.PM_FLAG_NO_CAUSTIC)
    pane, st = gpu_ctx.render(None, None, rp)
    g_direct = golden("pane_direct" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w)
    assert np.array_equal(pane, g_direct)
    rp = gpu_ctx.params(w, h, seed)
    pane, st = gpu_ctx.render(gm, cm, rp)
    g_can = golden("pane_full_canonical" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w)
    g_ref = golden("pane_full_ref" + tag(w, h, n) + ".bin", np.uint32).reshape(h, w)
    g_stats = golden("pane_stats" + tag(w, h, n) + ".bin", np.uint64)
    assert np.array_equal(pane, g_can)
    assert psnr(pane, g_ref) >= 40.0
    assert [st.closest_hit_queries, st.shadow_queries, st.gathers, st.absorb_points] == list(g_stats)
    gm.destroy(); cm.destroy()


def test_full_frame_bit_exact_vs_oracle_and_rank_split(gpu_ctx, orc):
    """GPU-emitted maps + GPU render == oracle render on the same photons; rendering the frame as two
    or three ranks' interleaved row blocks gives the identical pane."""
    w, h, n, seed = 320, 200, 30000, 99
    gm = gpu_ctx.create_map(0, n, seed)
    cm = gpu_ctx.create_map(1, n, seed)
    og, oc = orc.Map(gm.download()), orc.Map(cm.download())
    rp = gpu_ctx.params(w, h, seed)
    pane, st = gpu_ctx.render(gm, cm, rp)
    want, ost = orc.render(orc.default_scene(), og, oc, orc.params(w, h, seed), nthreads=8)
    assert np.array_equal(pane, want)
    assert (st.closest_hit_queries, st.shadow_queries, st.gathers) == (ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
    for world in (2, 3):
        split = np.zeros_like(pane)
        for rank in range(world):
            gpu_ctx.render(gm, cm, gpu_ctx.params(w, h, seed, rank=rank, world=world), pane=split)
        assert np.array_equal(split, pane)
    gm.destroy(); cm.destroy(); og.close(); oc.close()


@pytest.mark.parametrize("maps", ["auto", "allgather", "replicate"])
def test_bench_collective_path_on_one_rank(tmp_path, orc, maps):
    """bench.py through its distributed code path (RCCL all_gather of device photon shards -- or, with --maps replicate,
    every map emitted locally -- and the gather of row blocks) with a one-rank process group: same pane as the oracle."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "pane.bin"
    env = dict(os.environ, PM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--width", "320", "--height", "200",
                        "--photons", "20000", "--seed", "77", "--no-cpu-baseline", "--maps", maps, "--dump-pane", str(out)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["kernel"] and j["roofline"]["traffic"] is None   # (no counter passes beside a process group)
    tune = j["config"]["maps_autotune_ms_per_step"]
    assert (set(tune) == {"allgather", "replicate"}) if maps == "auto" else tune is None
    pane = np.fromfile(out, dtype=np.uint32).reshape(200, 320)
    scene = orc.default_scene()
    g, c = orc.Map(orc.emit(scene, 0, 77, 20000)[0]), orc.Map(orc.emit(scene, 1, 77, 20000)[0])
    want, _ = orc.render(scene, g, c, orc.params(320, 200, 77), nthreads=8)
    assert np.array_equal(pane, want)
    g.close(); c.close()


def test_bench_two_ranks_over_rccl(tmp_path, orc):
    """The real N > 1 path (emission split by attempt range, RCCL all-gather of the shards, row blocks gathered to rank 0):
    bench.py under torch.distributed.run with two ranks, its assembled pane against the oracle's.  Needs two GPUs."""
    import os, subprocess, sys, json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL path needs two")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for maps in ("allgather", "replicate"):
        out = tmp_path / f"pane_{maps}.bin"
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                            os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "200", "--photons", "20000",
                            "--seed", "77", "--no-cpu-baseline", "--maps", maps, "--dump-pane", str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["n_gpus"] == 2 and j["value"] > 0
        pane = np.fromfile(out, dtype=np.uint32).reshape(200, 320)
        scene = orc.default_scene()
        g, c = orc.Map(orc.emit(scene, 0, 77, 20000)[0]), orc.Map(orc.emit(scene, 1, 77, 20000)[0])
        want, _ = orc.render(scene, g, c, orc.params(320, 200, 77), nthreads=8)
        assert np.array_equal(pane, want), maps
        g.close(); c.close()
