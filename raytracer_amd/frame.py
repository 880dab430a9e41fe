"""Frame orchestration above the C-ABI: the host-side mirror of renderScene (reference
src/Raytrace.cpp:301-343) for one process per GPU.

Sharding: emission attempts are split into contiguous per-rank ranges and the
per-rank photon shards are combined with ONE all-gather (RCCL over xGMI when the process group is
NCCL); every rank then builds the same full map ("first n successes in global attempt order", so the
map does not depend on the number of GPUs).  Image rows are split into interleaved 16-row blocks;
row blocks are gathered to rank 0.  Without a process group (dist=None, single GPU) no collective is issued.

The compute is injected (``emit_fn`` etc.) so that the sharding logic can be exercised on CPU with the
gloo backend in tests; the product wiring (``GpuFrame``) always calls the HIP library.
"""
import math

import numpy as np

ROW_BLOCK = 16
YIELD_EST = {0: 0.60, 1: 0.36}   # stored photons per emission attempt, default scene (global / caustic)


def attempt_ranges(begin, total, world):
    """Contiguous per-rank attempt ranges covering [begin, begin + per*world): [(b0, per), ...]."""
    per = (total + world - 1) // world
    return [(begin + r * per, per) for r in range(world)]


def batch_attempts(want, kind, world, yields=YIELD_EST):
    return int(want / yields[kind]) + 4096 * world


def owned_row_blocks(height, rank, world):
    nblk = (height + ROW_BLOCK - 1) // ROW_BLOCK
    return list(range(rank, nblk, world))


def emit_sharded(n, kind, world, rank, emit_fn, allgather_fn, yields=YIELD_EST):
    """First ``n`` stored photons in global attempt order.

    emit_fn(begin, count) -> records of this rank's range in attempt order (array-like, rows = photons).
    allgather_fn(records) -> list of the per-rank record arrays, in rank order.
    Returns (list of record chunks in final order, attempts issued).
    """
    chunks, have, next_attempt = [], 0, 0
    empty_rounds = 0
    while have < n:
        want = n - have
        ranges = attempt_ranges(next_attempt, batch_attempts(want, kind, world, yields), world)
        mine = emit_fn(*ranges[rank])
        shards = allgather_fn(mine)
        got = 0
        for s in shards:                       # rank order == attempt order
            take = min(len(s), n - have)
            if take > 0:
                chunks.append(s[:take])
                have += take
                got += take
        next_attempt = ranges[-1][0] + ranges[-1][1]
        empty_rounds = empty_rounds + 1 if got == 0 else 0
        if empty_rounds > 8:
            raise RuntimeError("no emission attempt stores a photon in this scene")
    return chunks, next_attempt


def allgather_records(dist, mine, world):
    """The one exchange step of the path: per-rank photon shards ([n_r, 8] int32 rows = 32-byte records,
    any device) -> list of all ranks' shards in rank order.  Shards are padded to the largest count."""
    import torch
    if dist is None:
        return [mine]
    cnt = torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device)
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cap = max(int(c.item()) for c in cnts)
    pad = torch.zeros((cap, 8), dtype=torch.int32, device=mine.device)
    pad[: mine.shape[0]] = mine
    shards = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(shards, pad)
    return [s[: int(c.item())] for s, c in zip(shards, cnts)]


def gather_row_blocks(dist, pane, rank, world, cache=None):
    """Interleaved 16-row blocks of every rank's [H, W] int32 pane -> the full pane on rank 0 (None elsewhere).

    cache: a dict that keeps the staging buffers from frame to frame (GpuFrame passes its own).  With it, and a height that is
    a whole number of row blocks, a frame's gather is one packing copy, the collective and one unpacking copy, no allocation and
    no fill (the one-rank step through RCCL with the shard all-gather: 2.80 -> 2.75 ms); the pane returned is then a view of a cached buffer
    that the next frame overwrites."""
    import torch
    if dist is None:
        return pane
    h, w = pane.shape
    nblk = (h + ROW_BLOCK - 1) // ROW_BLOCK
    hp = nblk * ROW_BLOCK
    per = (nblk + world - 1) // world
    if cache is not None and hp == h:
        key = ("row_blocks", h, w, world, rank, str(pane.device))
        bufs = cache.get(key)
        if bufs is None:
            mine = torch.zeros((per, ROW_BLOCK, w), dtype=torch.int32, device=pane.device)
            every = torch.zeros((world, per, ROW_BLOCK, w), dtype=torch.int32, device=pane.device) if rank == 0 else None
            full = torch.zeros((per, world, ROW_BLOCK, w), dtype=torch.int32, device=pane.device) if rank == 0 else None
            bufs = cache[key] = (mine, every, full, list(every.unbind(0)) if rank == 0 else None)
        mine, every, full, parts = bufs
        own = pane.view(nblk, ROW_BLOCK, w)[rank::world]
        mine[: own.shape[0]].copy_(own)
        dist.gather(mine, parts, dst=0)
        if rank != 0:
            return None
        full.copy_(every.permute(1, 0, 2, 3))          # block i * world + r of the pane is block i of rank r
        return full.view(per * world * ROW_BLOCK, w)[:h]
    mine = torch.zeros((per, ROW_BLOCK, w), dtype=torch.int32, device=pane.device)
    padded = pane if hp == h else torch.cat([pane, torch.zeros((hp - h, w), dtype=torch.int32, device=pane.device)])
    own = padded.view(nblk, ROW_BLOCK, w)[rank::world]
    mine[: own.shape[0]] = own
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank != 0:
        return None
    full = torch.zeros((nblk, ROW_BLOCK, w), dtype=torch.int32, device=pane.device)
    for r in range(world):
        k = len(range(r, nblk, world))
        full[r::world] = parts[r][:k]
    return full.view(hp, w)[:h]


class GpuFrame:
    """renderScene on one rank of a (possibly single-rank) job; all compute through the HIP library."""

    def __init__(self, ctx, width, height, n_photons, seed, rank=0, world=1, dist=None,
                 cam=(0.0, 0.0, -12.0), light_color=(0.6, 0.6, 0.6), flags=0, replicate_maps=False):
        import torch
        self.torch, self.ctx, self.dist = torch, ctx, dist
        self.w, self.h, self.n, self.seed = width, height, n_photons, seed
        self.rank, self.world = rank, world
        self.cam, self.light_color, self.flags = cam, light_color, flags
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.pane = torch.zeros((height, width), dtype=torch.int32, device=self.dev)
        self.maps = [None, None]
        self.last_stats = None
        self._shard_bufs = {}       # kind -> (own shard, all shards), device byte tensors reused across frames
        self._gather_bufs = {}      # staging buffers of the pane gather, reused across frames
        self._fast_shards = dist is not None and hasattr(dist, "all_gather_into_tensor")
        self._yields = None         # stored photons per attempt in THIS scene, probed once (every rank finds the same numbers)
        # Every rank emits every map itself and nothing is exchanged (the random streams are counter-based: all ranks get the
        # same maps).  Not the default: the path as specified splits the emission and all-gathers the shards.  Which of the
        # two is faster depends on what the all-gather sustains against 5 G photons/s of emission per GPU.
        self.replicate_maps = replicate_maps

    # --- photon maps: createPhotonMap / createCausticPhotonMap, src/Raytrace.cpp:314-316
    def _emit_fn(self, kind):
        torch = self.torch

        def fn(begin, count):
            shard = torch.empty((count, 8), dtype=torch.int32, device=self.dev)
            got = self.ctx.emit_range(kind, self.seed, self.light_color, begin, count, shard.data_ptr(), count)
            return shard[:got]
        return fn

    def _allgather(self, mine):
        return allgather_records(self.dist, mine, self.world)

    PROBE_ATTEMPTS = 65536

    def yields(self):
        """Stored photons per emission attempt of the uploaded scene, from a short synchronous emission per map kind
        (attempts 0 .. 65535, the same on every rank), less a twentieth: sizes the ranks' attempt ranges."""
        if self._yields is None:
            torch = self.torch
            tmp = torch.empty((self.PROBE_ATTEMPTS, 8), dtype=torch.int32, device=self.dev)
            self._yields = {}
            for kind in (0, 1):
                got = self.ctx.emit_range(kind, self.seed, self.light_color, 0, self.PROBE_ATTEMPTS, tmp.data_ptr(), self.PROBE_ATTEMPTS)
                self._yields[kind] = max(0.95 * got / self.PROBE_ATTEMPTS, 1e-4)
        return self._yields

    def _build_map_from_shards(self, kind):
        """The sharded build without a host wait: this rank's attempt range -> a shard (header + records) on the
        device, ONE all-gather of the shards, assembly + index build queued behind it.  The ranges are sized so that
        together they hold n photons with room to spare; if they do not after all, the first use of the map raises and
        ``render`` falls back to ``emit_sharded`` (rounds of ranges with the counts on the host)."""
        torch = self.torch
        y = self.yields()
        per = (batch_attempts(self.n, kind, self.world, y) + self.world - 1) // self.world
        cap = int(per * min(1.0, y[kind] / 0.95 + 0.10)) + 1024
        nbytes = 32 + 32 * cap
        bufs = self._shard_bufs.get(kind)
        if bufs is None or bufs[0].numel() != nbytes:
            bufs = (torch.empty(nbytes, dtype=torch.uint8, device=self.dev), torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.dev))
            self._shard_bufs[kind] = bufs
        mine, every = bufs
        self.ctx.emit_shard(kind, self.seed, self.light_color, self.rank * per, per, mine.data_ptr(), cap)
        self.dist.all_gather_into_tensor(every, mine)
        return self.ctx.build_map_shards(every.data_ptr(), self.world, nbytes, kind, self.n)

    def _build_map_by_rounds(self, kind):
        torch = self.torch
        chunks, _ = emit_sharded(self.n, kind, self.world, self.rank, self._emit_fn(kind), self._allgather, self.yields())
        rec = torch.cat(chunks) if len(chunks) != 1 else chunks[0].contiguous()
        return self.ctx.build_map(rec.data_ptr(), kind, n=self.n, on_device=True)

    def build_maps(self, fast=True):
        for m in self.maps:
            if m is not None:
                m.destroy()
        self.maps = [None, None]
        for kind in (1, 0):   # the caustic map takes longer (more attempts per photon): its build goes first
            if self.flags & (1 << kind):
                continue
            if self.dist is None or self.replicate_maps:
                self.maps[kind] = self.ctx.create_map(kind, self.n, self.seed, self.light_color, wait=False)
            elif fast and self._fast_shards:
                self.maps[kind] = self._build_map_from_shards(kind)
            else:
                self.maps[kind] = self._build_map_by_rounds(kind)

    # --- the job loop + waitForCompletion, src/Raytrace.cpp:322-336
    def render(self):
        from .capi import PMError
        rp = self.ctx.params(self.w, self.h, self.seed, self.cam, self.light_color, self.flags, self.rank, self.world)
        try:
            _, st = self.ctx.render(self.maps[0], self.maps[1], rp, d_pane=self.pane.data_ptr())
        except PMError as e:
            if self.dist is None or "pm_map_build_shards" not in str(e):
                raise
            # the shards did not hold n photons (every rank sees the same counts, so every rank lands here)
            self._fast_shards = False
            self.build_maps(fast=False)
            _, st = self.ctx.render(self.maps[0], self.maps[1], rp, d_pane=self.pane.data_ptr())
        self.last_stats = st
        return st

    def gather_pane(self):
        """Row blocks of all ranks -> rank 0's full pane (returns it on rank 0, None elsewhere)."""
        return gather_row_blocks(self.dist, self.pane, self.rank, self.world, self._gather_bufs)

    def step(self):
        """One whole frame: both photon maps, this rank's rows, pane on rank 0."""
        self.build_maps()
        st = self.render()
        pane = self.gather_pane()
        return st, pane

    def close(self):
        for m in self.maps:
            if m is not None:
                m.destroy()
        self.maps = [None, None]
