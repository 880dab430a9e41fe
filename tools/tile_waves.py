"""Diagnostic (a build with -DPM_WAVE_TIMES, PM_LIB=...): when do the waves of the group kernels start and stop within a step?
Per map: the launch's span, how much of the resident waves' time lies after their last group (idle end), groups per wave."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra
from raytracer_amd.frame import GpuFrame
lib = ra.load_library()
ctx = ra.Context(0)
scene = {"default": ra.default_scene, "spheres10k": ra.many_spheres_scene, "caustics": ra.caustics_scene}[sys.argv[1] if len(sys.argv) > 1 else "default"]()
ctx.upload_scene(scene)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # > 1: rank 0's share of the row blocks of a world of that size
if world == 1:
    frame = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
    step = frame.step
else:
    pane = torch.zeros((1440, 1440), dtype=torch.int32, device="cuda")
    def step():
        g = ctx.create_map(0, 1000000, 12345, wait=False); c = ctx.create_map(1, 1000000, 12345, wait=False)
        rp = ctx.params(1440, 1440, 12345); rp.rank = 0; rp.world = world
        _, st = ctx.render(g, c, rp, d_pane=pane.data_ptr())
        torch.cuda.synchronize()
        print("   ", {k: round(getattr(st, k), 3) for k in ("ms_shadow", "ms_gather", "ms_gather_global", "ms_gather_caustic", "ms_group_global", "ms_group_caustic", "ms_total")}, [tuple(round(x, 3) for x in m.times()) for m in (g, c)])
        g.destroy(); c.destroy()
for _ in range(3):
    step()
torch.cuda.synchronize()
raw = np.zeros(2 * 8192 * 4 + 2 * 16384 * 2, dtype=np.uint32)
lib.pm_tile_waves(None, 1)
step(); torch.cuda.synchronize()
lib.pm_tile_waves(raw.ctypes.data_as(ctypes.c_void_p), 0)
buf = raw[:2 * 8192 * 4].reshape(2, 8192, 4); grp = raw[2 * 8192 * 4:].reshape(2, 16384, 2)
for kind, name in enumerate(("global", "caustic")):
    a = buf[kind]
    used = a[:, 2] != 0 if world > 1 else a[:, 0] != 0         # (a small share: only the waves that had a group)
    t0, t1, ng, xcd, blk = a[used, 0].astype(np.int64), a[used, 1].astype(np.int64), a[used, 2], a[used, 3] >> 8, a[used, 3] & 255
    print('   blockIdx & 7 against the XCC_ID register:', [sorted(set(xcd[blk == b].tolist())) for b in range(8)])
    s = t0 - t0.min(); e = t1 - t0.min()      # ticks of the device-wide 100 MHz clock
    span = e.max()
    print(f"{name}: {used.sum()} waves ran, span {span} (x 10 ns); start p50 {np.median(s)} p99 {np.percentile(s, 99):.0f}; last group ends: p10 {np.percentile(e,10):.0f} p50 {np.median(e):.0f} p90 {np.percentile(e,90):.0f} max {span}")
    print(f"   groups per wave: mean {ng.mean():.2f} min {ng.min()} max {ng.max()}; waves without a group {(ng == 0).sum()}")
    print(f"   slot time after the last group / span x waves: {(span - e).sum() / (span * len(e)):.3f}; before the first: {s.sum() / (span * len(e)):.3f}")
    early = s < 0.05 * span
    he, _ = np.histogram(e[early] / span, bins=10, range=(0, 1))
    print(f"   {early.sum()} waves start in the first twentieth of the span, {ng[early].sum()} of {ng.sum()} groups are theirs; they finish, by tenth of the span: {he.tolist()}")
    print(f"   their slot time after the last group: {(span - e[early]).sum() / (span * early.sum()):.3f} of span x waves")
    hist, _ = np.histogram(e / span, bins=10, range=(0, 1))
    print("   waves by the tenth of the span in which they finish:", hist.tolist())
    g = grp[kind]; ok = g[:, 1] != 0
    dur = (g[ok, 1].astype(np.int64) - g[ok, 0].astype(np.int64)) / 100.0      # us
    st = (g[ok, 0].astype(np.int64) - t0.min()) / 100.0
    idx = np.nonzero(ok)[0]
    print(f"   {ok.sum()} groups: us per group mean {dur.mean():.0f} p50 {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} p99 {np.percentile(dur, 99):.0f} max {dur.max():.0f}; sum {dur.sum() / 1e3:.1f} wave-ms")
    print("   mean us per group by 512 groups of the draw order:", [int(dur[(idx >= b) & (idx < b + 512)].mean()) for b in range(0, int(idx.max()) + 1, 512)])
    print("   mean start (us) by 512 groups:", [int(st[(idx >= b) & (idx < b + 512)].mean()) for b in range(0, int(idx.max()) + 1, 512)])
    print("   by XCD (blockIdx & 7): groups", [int(ng[blk == b].sum()) for b in range(8)], "last end (us)", [int(e[blk == b].max() / 100) for b in range(8)], "waves with work", [int((ng[blk == b] > 0).sum()) for b in range(8)])
    late = st + dur > 0.85 * (span / 100.0)
    print(f"   groups still running after 85 % of the span: {late.sum()}, their mean duration {dur[late].mean():.0f} us, started at {st[late].mean():.0f} us on average")
