"""CPU tests of the N > 1 path: attempt-range sharding + all-gather assembly of the photon map and the
interleaved row-block gather of the pane, with the oracle standing in for the GPU kernels.  The
multi-process tests run the product's own collective helpers (raytracer_amd.frame) on gloo, world 2 and world 8."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import frame


def _emit_fn(orc, kind, seed):
    scene = orc.default_scene()

    def fn(begin, count):
        out = []
        for a in range(begin, begin + count):
            ok, ph, _ = orc.emit_attempt(scene, kind, seed, a)
            if ok:
                out.append(ph)
        return np.array(out, dtype=orc.photon_dtype) if out else np.zeros(0, dtype=orc.photon_dtype)
    return fn


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", [0, 1])
def test_sharded_emission_is_independent_of_world_size(orc, world, kind):
    """First n stored photons in global attempt order, whatever the number of ranks."""
    n, seed = 700, 31
    want, _ = orc.emit(orc.default_scene(), kind, seed, n)
    fns = [_emit_fn(orc, kind, seed) for _ in range(world)]
    # simulate the ranks in lock-step: the "all-gather" returns every rank's shard for the same round
    results = []
    for rank in range(world):
        def allgather(mine, _rank=rank):
            return [fns[r](*cur[r]) for r in range(world)]
        cur = None
        orig = frame.attempt_ranges

        def spy(begin, total, w):
            nonlocal cur
            cur = orig(begin, total, w)
            return cur
        frame.attempt_ranges = spy
        try:
            chunks, _ = frame.emit_sharded(n, kind, world, rank, fns[rank], allgather)
        finally:
            frame.attempt_ranges = orig
        results.append(np.concatenate(chunks))
    for r in results:
        assert r.tobytes() == want.tobytes()


def test_row_block_ownership():
    for h in (1, 15, 16, 17, 1440, 90):
        for world in (1, 2, 3, 8):
            seen = sorted(b for r in range(world) for b in frame.owned_row_blocks(h, r, world))
            assert seen == list(range((h + 15) // 16))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _row_marks(h, w):
    y, x = np.mgrid[0:h, 0:w]
    return ((y.astype(np.uint64) * 2654435761 + x.astype(np.uint64) * 40503) & 0xFFFFFFFF).astype(np.uint32)


def _worker(rank, world, port, w, h, n, seed, q):
    import torch
    import torch.distributed as dist
    from oracle import orc_py as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = orc.default_scene()
    maps = []
    for kind in (0, 1):
        fn = _emit_fn(orc, kind, seed)

        def emit_t(begin, count):
            return torch.from_numpy(fn(begin, count).view(np.int32).reshape(-1, 8).copy())
        chunks, _ = frame.emit_sharded(n, kind, world, rank, emit_t, lambda mine: frame.allgather_records(dist, mine, world))
        rec = torch.cat(chunks).numpy().view(orc.photon_dtype).reshape(-1)
        maps.append(rec)
    g, c = orc.Map(maps[0]), orc.Map(maps[1])
    pane = np.zeros((h, w), dtype=np.uint32)
    rp = orc.params(w, h, seed)
    for b in frame.owned_row_blocks(h, rank, world):
        orc.render(scene, g, c, rp, y0=b * 16, y1=min(h, b * 16 + 16), pane=pane)
    full = frame.gather_row_blocks(dist, torch.from_numpy(pane.view(np.int32)), rank, world)
    # the same gather on a pane whose every row is different (most rows of a tall frame miss the scene and are one colour):
    # a row block that landed in the wrong place would show here
    marks = np.zeros((h, w), dtype=np.uint32)
    for b in frame.owned_row_blocks(h, rank, world):
        marks[b * 16:min(h, b * 16 + 16)] = _row_marks(h, w)[b * 16:min(h, b * 16 + 16)]
    # through the cached staging buffers GpuFrame uses (whole row blocks only: the other heights take the plain path), twice: the
    # second frame reuses the first one's buffers
    cache = {}
    first = frame.gather_row_blocks(dist, torch.from_numpy((marks ^ np.uint32(0x5A5A)).view(np.int32)), rank, world, cache)
    if rank == 0:
        assert np.array_equal(first.numpy().view(np.uint32), _row_marks(h, w) ^ np.uint32(0x5A5A))
    full_marks = frame.gather_row_blocks(dist, torch.from_numpy(marks.view(np.int32)), rank, world, cache)
    assert (len(cache) == 1) == (h % 16 == 0)
    if rank == 0:
        q.put((maps[0].tobytes(), maps[1].tobytes(), full.numpy().view(np.uint32).copy(), full_marks.numpy().view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 96, 70),        # 70 rows: 5 row blocks, the last one partial, uneven split
                                       (8, 128, 1440)])    # the bench frame's 1440 rows: 90 row blocks over 8 ranks, 11 or 12 each
def test_multi_rank_frame_equals_single_rank(orc, world, w, h):
    """gloo, world_size 2 and 8: photon maps and the gathered pane are identical to the 1-rank result."""
    import torch.multiprocessing as mp
    n, seed = 1500, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    g_bytes, c_bytes, pane, marks = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene = orc.default_scene()
    wg, _ = orc.emit(scene, 0, seed, n)
    wc, _ = orc.emit(scene, 1, seed, n)
    assert g_bytes == wg.tobytes() and c_bytes == wc.tobytes()
    g, c = orc.Map(wg), orc.Map(wc)
    want, _ = orc.render(scene, g, c, orc.params(w, h, seed), nthreads=4)
    assert np.array_equal(pane, want)
    assert np.array_equal(marks, _row_marks(h, w))
    g.close(); c.close()
