"""Randomised sweep of the standalone gather entry point (pm_knn_gather) and of pm_knn_query against the oracle: random maps
(both kinds, 1 .. 400 000 photons) and query points anywhere -- on photons, on surfaces, in mid air, far outside the scene,
with arbitrary normals.  usage: fuzz_knn.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from oracle import orc_py as orc


def sweep(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = ra.Context(0)
    ctx.upload_scene(ra.default_scene())
    bad = []
    t0 = time.time()
    try:
        for k in range(cases):
            kind = int(rng.integers(0, 2))
            n = int(rng.choice([1, 62, 63, 64, 200, 5000, 50000, 400000]))
            sd = int(rng.integers(0, 2 ** 40))
            m = ctx.create_map(kind, n, sd)
            ph = m.download()
            om = orc.Map(ph)
            npts = 300
            on_photon = np.stack([ph["x"], ph["y"], ph["z"]], 1)[rng.integers(0, n, npts)].astype(np.float64)
            jitter = on_photon + rng.normal(0, 0.05, (npts, 3))
            air = rng.uniform([-5, -5, 0], [5, 5, 10], (npts, 3))
            far = rng.uniform(-30, 30, (npts // 4, 3))
            pts = np.concatenate([on_photon[: npts // 4], jitter, air, far])
            nrm = rng.normal(0, 1, pts.shape)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            axis = np.eye(3)[rng.integers(0, 3, len(pts))] * rng.choice([-1.0, 1.0], (len(pts), 1))
            nrm = np.where(rng.random((len(pts), 1)) < 0.5, axis, nrm)        # half of them axis-aligned, as on the walls
            rad = ctx.knn_gather(m, pts, nrm)
            want = om.gather(kind, pts, nrm, order=orc.ORDER_CANONICAL, nthreads=16)
            ok = np.array_equal(rad.view(np.uint64), want.view(np.uint64))
            found, idx, d2 = ctx.knn_query(m, pts[:40])
            for q in range(40):
                f, oi, od = om.knn_exact(pts[q])
                ok = ok and found[q] == f and np.array_equal(d2[q, :f].view(np.uint64), np.asarray(od)[:f].view(np.uint64))
            if not ok:
                bad.append(f"case {k}: kind {kind} n={n} seed={sd} differing radiance rows={int((rad.view(np.uint64) != want.view(np.uint64)).any(1).sum())}")
                if verbose:
                    print("MISMATCH", bad[-1], flush=True)
            m.destroy(); om.close()
            if verbose and k % 20 == 19:
                print(f"{k + 1} cases, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
    finally:
        ctx.close()
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    found_bad = sweep(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"done: {n_cases} cases, {len(found_bad)} mismatches")
    sys.exit(1 if found_bad else 0)
