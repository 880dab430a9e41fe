"""GPU tests of the two gather paths and of the asynchronous map construction, through the C-ABI.

The group gather (one lane per query, pm_tile.h) and the per-query gather (one wave per query, pm_grid.h)
must give the same bits; a map whose emission and build are left on a build stream must be the map the
waiting call produces; an emission range that falls short of n photons must be repaired.
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(monkeypatch, tile, w, h, n, seed, flags=0):
    import raytracer_amd as ra
    monkeypatch.setenv("PM_GATHER_TILE", "1" if tile else "0")   # read once per context, at its first gather
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(ra.default_scene())
        g, c = ctx.create_map(0, n, seed), ctx.create_map(1, n, seed)
        rp = ctx.params(w, h, seed)
        rp.flags = flags
        pane, st = ctx.render(g, c, rp)
        g.destroy(); c.destroy()
        return pane, st
    finally:
        ctx.close()


def test_group_gather_matches_per_query_gather(monkeypatch):
    """Dense maps (most tiles are gathered as groups): identical panes with and without the group kernel."""
    w, h, n, seed = 640, 400, 200000, 4242
    pane_t, st_t = _render(monkeypatch, True, w, h, n, seed)
    pane_q, st_q = _render(monkeypatch, False, w, h, n, seed)
    assert np.array_equal(pane_t, pane_q)
    assert st_q.grouped_global == 0 and st_q.grouped_caustic == 0
    # the group kernel really served most of the primary shade points
    assert st_t.grouped_global > 0.5 * st_t.absorb_points and st_t.grouped_caustic > 0.3 * st_t.absorb_points
    assert (st_t.gathers, st_t.shadow_queries) == (st_q.gathers, st_q.shadow_queries)


def test_ordering_of_bounced_points_beside_or_behind_the_shadow_pass(monkeypatch, orc):
    """The bounced pixels' shade points are put in spatial order on a stream of their own beside the shadow pass (the default)
    or behind it on the caller's stream (PM_SP_STREAM=0, read when the context is created): the same pane, and the oracle's."""
    w, h, n, seed = 400, 260, 60000, 99
    monkeypatch.setenv("PM_SP_STREAM", "1")
    pane_side, st_side = _render(monkeypatch, True, w, h, n, seed)
    monkeypatch.setenv("PM_SP_STREAM", "0")
    pane_behind, st_behind = _render(monkeypatch, True, w, h, n, seed)
    assert np.array_equal(pane_side, pane_behind)
    assert (st_side.gathers, st_side.grouped_global, st_side.grouped_caustic) == (st_behind.gathers, st_behind.grouped_global, st_behind.grouped_caustic)
    scene = orc.default_scene()
    og, oc = orc.Map(orc.emit(scene, 0, seed, n)[0]), orc.Map(orc.emit(scene, 1, seed, n)[0])
    want, _ = orc.render(scene, og, oc, orc.params(w, h, seed), nthreads=16)
    og.close(); oc.close()
    assert np.array_equal(pane_side, want)


@pytest.mark.parametrize("n", [40, 3000])
def test_group_gather_sparse_maps(monkeypatch, n):
    """Maps so sparse that groups are handed over (fewer than 63 photons, radius beyond MAX_R2, huge spread)."""
    pane_t, _ = _render(monkeypatch, True, 320, 200, n, 7)
    pane_q, _ = _render(monkeypatch, False, 320, 200, n, 7)
    assert np.array_equal(pane_t, pane_q)


def test_async_map_creation_is_the_same_map(gpu_ctx):
    """wait=False leaves emission + build on a build stream; the first use waits.  Same photons, same pane."""
    n, seed = 50000, 31
    a = [gpu_ctx.create_map(k, n, seed, wait=False) for k in (1, 0)]     # both in flight at once
    b = [gpu_ctx.create_map(k, n, seed, wait=True) for k in (1, 0)]
    rp = gpu_ctx.params(256, 160, seed)
    pane_a, _ = gpu_ctx.render(a[1], a[0], rp)                            # render is the first use
    pane_b, _ = gpu_ctx.render(b[1], b[0], rp)
    assert np.array_equal(pane_a, pane_b)
    for ma, mb in zip(a, b):
        assert np.array_equal(ma.download().view(np.uint8), mb.download().view(np.uint8))
    for m in a + b:
        m.destroy()


@pytest.mark.parametrize("kind", [0, 1])
def test_ragged_attempt_ranges_and_both_emission_kernels(orc, kind):
    """pm_emit_range on ranges that end in the middle of a wave, of a run of 256 attempts and of a block of 1 024: the
    stored photons are the oracle's for exactly those attempts, whichever kernel emits them (lanes refilled from a wave's run
    of attempts: the default; one attempt per lane: PM_EMIT_RUN=0, read when the scene is uploaded)."""
    import torch
    import raytracer_amd as ra
    seed, lc = 77, (0.6, 0.6, 0.6)
    scene = orc.default_scene()
    kept = {}
    for env in ("1", "0"):
        os.environ["PM_EMIT_RUN"] = env
        ctx = ra.Context(0)
        try:
            ctx.upload_scene(ra.default_scene())
            for begin, count in ((0, 1), (5, 63), (1000, 64), (123, 65), (0, 255), (9, 257), (4096, 1023), (77, 1025), (31, 5000)):
                out = torch.zeros((count, 8), dtype=torch.int32, device="cuda")
                got = ctx.emit_range(kind, seed, lc, begin, count, out.data_ptr(), count)
                rec = out.cpu().numpy()[:got].copy().view(ra.photon_dtype).reshape(-1)
                want = [ph for ok, ph, _ in (orc.emit_attempt(scene, kind, seed, a) for a in range(begin, begin + count)) if ok]
                assert got == len(want), (env, begin, count)
                assert rec.tobytes() == (np.array(want, dtype=ra.photon_dtype).tobytes() if want else b""), (env, begin, count)
                if env == "1": kept[(begin, count)] = rec.tobytes()
                else: assert kept[(begin, count)] == rec.tobytes()
        finally:
            ctx.close()
            os.environ.pop("PM_EMIT_RUN", None)


def test_emission_shortfall_is_repaired():
    """A scene that stores far fewer photons per attempt than the reference's (floor only): the one emission range the
    library sizes for n photons falls short, and the map must still be the first n successes in attempt order --
    here checked against ranges cut differently through pm_emit_range."""
    import raytracer_amd as ra
    hip = ctypes.CDLL("libamdhip64.so")
    scene = ra.default_scene()[:1].copy()       # the floor plane alone: most attempts leave the scene
    n, seed, kind, lc = 20000, 5, 0, (0.6, 0.6, 0.6)
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(scene)
        m = ctx.create_map(kind, n, seed, lc, wait=False)
        got = m.download()
        assert len(got) == n
        # the same photons from attempt ranges of 7001 attempts each
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), n * 32) == 0
        have, a0, step = 0, 0, 7001
        while have < n:
            cnt = ctx.emit_range(kind, seed, lc, a0, step, d.value + have * 32, n - have)
            have += min(cnt, n - have)
            a0 += step
        assert a0 > n / 0.60 + 4096, "the scene was meant to make the library's first range fall short"
        ref = ctx.build_map(d.value, kind, n=n, on_device=True)
        want = ref.download()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        # and the repaired map answers queries like the reference-built one
        pts = np.stack([want["x"][:200], want["y"][:200] + 0.05, want["z"][:200]], 1).astype(np.float64)
        fa, ia, da = ctx.knn_query(m, pts)
        fb, ib, db = ctx.knn_query(ref, pts)
        assert np.array_equal(fa, fb) and np.array_equal(da.view(np.uint64), db.view(np.uint64)) and np.array_equal(ia, ib)
        m.destroy(); ref.destroy()
        hip.hipFree(d)
    finally:
        ctx.close()


def _shards(ctx, hip, kind, n, seed, world, per, cap):
    """`world` ranks' shards of attempt ranges of `per` attempts, laid out as the all-gather would (one device buffer)."""
    stride = 32 + 32 * cap
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), stride * world) == 0
    for r in range(world):
        ctx.emit_shard(kind, seed, (0.6, 0.6, 0.6), r * per, per, d.value + r * stride, cap)
    return d, stride


@pytest.mark.parametrize("kind", [0, 1])
def test_map_from_shards_is_the_single_rank_map(gpu_ctx, kind):
    """Three ranks' attempt ranges, emitted without a wait and assembled on the device, give the very map one rank
    emits alone (the map does not depend on the number of GPUs)."""
    hip = ctypes.CDLL("libamdhip64.so")
    n, seed, world = 30000, 11, 3
    per = (int(n / (0.60 if kind == 0 else 0.36)) + 4096 * world + world - 1) // world
    cap = int(per * ((0.60 if kind == 0 else 0.36) + 0.10)) + 1024
    d, stride = _shards(gpu_ctx, hip, kind, n, seed, world, per, cap)
    m = gpu_ctx.build_map_shards(d.value, world, stride, kind, n)
    one = gpu_ctx.create_map(kind, n, seed)
    assert np.array_equal(m.download().view(np.uint8), one.download().view(np.uint8))
    pts = np.random.default_rng(3).uniform([-5, -5, 0], [5, 5, 10], size=(300, 3))
    fa, ia, da = gpu_ctx.knn_query(m, pts)
    fb, ib, db = gpu_ctx.knn_query(one, pts)
    assert np.array_equal(fa, fb) and np.array_equal(ia, ib) and np.array_equal(da.view(np.uint64), db.view(np.uint64))
    m.destroy(); one.destroy(); hip.hipFree(d)


def test_map_from_shards_reports_shortfall_and_overflow(gpu_ctx):
    import raytracer_amd as ra
    hip = ctypes.CDLL("libamdhip64.so")
    # too few attempts for n photons
    d, stride = _shards(gpu_ctx, hip, 0, 10000, 3, 2, 2000, 4096)
    m = gpu_ctx.build_map_shards(d.value, 2, stride, 0, 10000)
    with pytest.raises(ra.PMError, match="fewer than n photons"):
        m.download()
    m.destroy(); hip.hipFree(d)
    # enough attempts, but shards too small to hold their successes
    d, stride = _shards(gpu_ctx, hip, 0, 10000, 3, 2, 12000, 1000)
    m = gpu_ctx.build_map_shards(d.value, 2, stride, 0, 10000)
    with pytest.raises(ra.PMError, match="overflowed"):
        m.download()
    m.destroy(); hip.hipFree(d)


def test_config3_size_group_gather_equals_per_query_gather(monkeypatch):
    """BASELINE config 3's sizes on one GPU (pane 7680 x 4320 = 3840 x 2160 at 4 spp, 10M + 10M photons): too large for the
    oracle, so the property that does not depend on size -- the group kernel and the per-query kernel are two exact
    searches and must agree bit for bit -- is checked on the whole pane, and the row split over two ranks on top."""
    import raytracer_amd as ra
    w, h, n, seed = 7680, 4320, 10_000_000, 12345
    panes = []
    for tile in (True, False):
        monkeypatch.setenv("PM_GATHER_TILE", "1" if tile else "0")
        ctx = ra.Context(0)
        try:
            ctx.upload_scene(ra.default_scene())
            g, c = ctx.create_map(0, n, seed, wait=False), ctx.create_map(1, n, seed, wait=False)
            pane, st = ctx.render(g, c, ctx.params(w, h, seed))
            if tile:
                assert st.grouped_global > 0.8 * st.absorb_points
                parts = np.zeros_like(pane)
                for rank in range(2):
                    ctx.render(g, c, ctx.params(w, h, seed, rank=rank, world=2), pane=parts)
                assert np.array_equal(parts, pane)
            panes.append(pane)
            g.destroy(); c.destroy()
        finally:
            ctx.close()
    assert np.array_equal(panes[0], panes[1])
    assert len(np.unique(panes[0])) > 1000


def test_config3_size_rows_equal_the_oracle():
    """BASELINE config 3's sizes again, against the oracle itself: the oracle cannot render 33 million pixels in a test, but
    it can render every 16th row of them on the same 2 x 10M photons (bench.py --check does exactly that: GPU maps
    downloaded, oracle index on them, 270 rows, 25 million rays, 1.9 million gathers).  Those rows are bit-identical."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--width", "7680", "--height", "4320",
                        "--photons", "10000000", "--cpu-row-stride", "16", "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["cpu_baseline"]["sample_rows_bit_identical_to_gpu"] is True
    assert "270 rows" in j["cpu_baseline"]["sample"]
