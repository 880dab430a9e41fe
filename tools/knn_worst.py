"""Diagnostics: the most expensive k-NN queries of a map on BASELINE config 2 shade points, with the photon
density around them (brute force on the host)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
rp = ctx.params(1440, 1440, 12345)
ids, pts, nrm = ctx.primary_hits(rp)
flat = ids.ravel()
sel = np.flatnonzero(flat >= 0)[::4]
m = ctx.create_map(kind, n, 12345)
found, idx, d2, work = ctx.knn_profile(m, pts[sel])
rec = m.download()
pos = np.stack([rec['x'], rec['y'], rec['z']], 1) if rec.dtype.names and 'x' in rec.dtype.names else None
w = work.astype(np.int64)
print("total leaves", w[:, 1].sum(), "mean", w[:, 1].mean(), "queries", len(sel))
order = np.argsort(-w[:, 1])
cum = np.cumsum(w[order, 1])
for frac in (0.001, 0.01, 0.1):
    k = int(len(sel) * frac)
    print(f"top {frac*100:.1f}% of queries hold {cum[k]/cum[-1]*100:.1f}% of the chunks")
print("bbox", None if pos is None else (pos.min(0), pos.max(0)))
for qi in order[:12]:
    p = pts[sel[qi]]
    r63 = d2[qi, :found[qi]].max() if found[qi] else -1
    line = f"obj {flat[sel[qi]]} p=({p[0]:.3f},{p[1]:.3f},{p[2]:.3f}) attempts={w[qi,0]} chunks={w[qi,1]} reduces={w[qi,2]} cands={w[qi,3]} found={found[qi]} r63^2={r63:.3e}"
    if pos is not None:
        dd = ((pos - p) ** 2).sum(1)
        line += " within(r63^2 x 1,4,16,64,256): " + ",".join(str(int((dd < r63 * f).sum())) for f in (1, 4, 16, 64, 256))
    print(line)
