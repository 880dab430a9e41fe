"""Diagnostics: per-query work counters of the k-NN traversal on BASELINE config 2 shade points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
rp = ctx.params(1440, 1440, 12345)
ids, pts, nrm = ctx.primary_hits(rp)
only = sys.argv[2] if len(sys.argv) > 2 else 'walls'
flat = ids.ravel()
sel = np.flatnonzero((flat >= 0) & (flat <= 4))[::8] if only == 'walls' else np.flatnonzero(flat >= 5)[::2]
for kind in (0, 1):
    m = ctx.create_map(kind, n, 12345)
    t = time.time(); found, idx, d2, work = ctx.knn_profile(m, pts[sel]); dt = time.time() - t
    fast = (work[:, 0] & 0x80000000) != 0          # the block query answered (no walk)
    work = work.copy(); work[:, 0] &= 0x7FFFFFFF
    w = work.astype(np.float64)
    print(f"   block query answered {fast.mean() * 100:.1f}% of the queries")
    print(f"kind {kind} n={n} queries={len(sel)} wall={dt*1e3:.1f}ms  mean nodes={w[:,0].mean():.1f} leaves={w[:,1].mean():.1f} reduces={w[:,2].mean():.2f} cands={w[:,3].mean():.1f}")
    for p in (50, 90, 99, 100):
        print(f"   p{p}: nodes={np.percentile(w[:,0],p):.0f} leaves={np.percentile(w[:,1],p):.0f} reduces={np.percentile(w[:,2],p):.0f} cands={np.percentile(w[:,3],p):.0f}")
    m.destroy()
