"""Randomised sweep of the multi-GPU map path on one GPU: `world` ranks' attempt ranges emitted as shards (pm_emit_shard),
laid out as the all-gather would, assembled and indexed (pm_map_build_shards) = the map one rank emits alone, for random
photon counts, worlds, seeds and both kinds; ranges cut by the frame's own rule (GpuFrame: probed yield).  Also ranges cut
at random through pm_emit_range.  usage: fuzz_shards.py [cases] [seed]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd.frame import batch_attempts


def sweep(cases, seed, verbose=True):
    hip = ctypes.CDLL("libamdhip64.so")
    rng = np.random.default_rng(seed)
    scenes = [ra.default_scene(), ra.caustics_scene(), ra.many_spheres_scene(300, 5)]
    bad = []
    t0 = time.time()
    for k in range(cases):
        scene = scenes[int(rng.integers(0, 3))]
        ctx = ra.Context(0)
        try:
            ctx.upload_scene(scene)
            kind, world = int(rng.integers(0, 2)), int(rng.integers(1, 9))
            n, sd = int(rng.choice([1, 100, 5000, 60000, 250000])), int(rng.integers(0, 2 ** 40))
            lc = (0.6, 0.6, 0.6)
            one = ctx.create_map(kind, n, sd, lc)
            want = one.download()
            # the frame's rule: a 65 536-attempt probe of the yield, ranges with room to spare
            probe = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(probe), 65536 * 32) == 0
            y = max(0.95 * ctx.emit_range(kind, sd, lc, 0, 65536, probe.value, 65536) / 65536, 1e-4)
            hip.hipFree(probe)
            per = (batch_attempts(n, kind, world, {kind: y}) + world - 1) // world
            cap = int(per * min(1.0, y / 0.95 + 0.10)) + 1024
            stride = 32 + 32 * cap
            d = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(d), stride * world) == 0
            for r in range(world):
                ctx.emit_shard(kind, sd, lc, r * per, per, d.value + r * stride, cap)
            m = ctx.build_map_shards(d.value, world, stride, kind, n)
            ok = np.array_equal(m.download().view(np.uint8), want.view(np.uint8))
            # random cuts through pm_emit_range
            d2 = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(d2), max(n, 1) * 32) == 0
            have, a0 = 0, 0
            while have < n:
                step = int(rng.integers(1, max(2, 3 * n)))
                cnt = ctx.emit_range(kind, sd, lc, a0, step, d2.value + have * 32, n - have)
                have += min(cnt, n - have); a0 += step
            m2 = ctx.build_map(d2.value, kind, n=n, on_device=True)
            ok = ok and np.array_equal(m2.download().view(np.uint8), want.view(np.uint8))
            pts = np.random.default_rng(k).uniform([-5, -5, 0], [5, 5, 10], (50, 3))
            fa, ia, da = ctx.knn_query(m, pts); fb, ib, db = ctx.knn_query(one, pts)
            ok = ok and np.array_equal(fa, fb) and np.array_equal(da.view(np.uint64), db.view(np.uint64))
            if not ok:
                bad.append(f"case {k}: kind {kind} world {world} n={n} seed={sd}")
                if verbose:
                    print("MISMATCH", bad[-1], flush=True)
            for x in (m, m2, one):
                x.destroy()
            hip.hipFree(d); hip.hipFree(d2)
        finally:
            ctx.close()
        if verbose and k % 20 == 19:
            print(f"{k + 1} cases, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    found = sweep(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"done: {n_cases} cases, {len(found)} mismatches")
    sys.exit(1 if found else 0)
