"""Diagnostics (needs a library built with -DPM_STAMPS, PM_LIB=...): where one per-query 63-NN search spends its time.
work words under PM_STAMPS = ticks: total, node sections (cell cover / splits), leaf sections (photon scans), reduces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
rp = ctx.params(1440, 1440, 12345)
ids, pts, nrm = ctx.primary_hits(rp)
flat = ids.ravel()
for name, sel in (("walls (every 16th hit)", np.flatnonzero((flat >= 0) & (flat < 5))[::16]), ("spheres (every 4th hit)", np.flatnonzero(flat >= 5)[::4])):
    for kind in (0, 1):
        m = ctx.create_map(kind, n, 12345)
        found, idx, d2, work = ctx.knn_profile(m, pts[sel])
        w = work.astype(np.float64)
        tot = w[:, 0].mean()
        print(f"{name:24s} map {kind}: {len(sel)} queries, ticks/query {tot:8.0f}: nodes {w[:,1].mean()/tot*100:4.1f}%  leaves {w[:,2].mean()/tot*100:4.1f}%  reduces {w[:,3].mean()/tot*100:4.1f}%  rest {(1-(w[:,1]+w[:,2]+w[:,3]).mean()/tot)*100:4.1f}%")
        m.destroy()
