"""Diagnostics: world-space extent of the 8x8 pixel tiles of BASELINE config 2 against the k-NN radius at their centre."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
rp = ctx.params(1440, 1440, 12345)
ids, pts, nrm = ctx.primary_hits(rp)
H = W = 1440
ids = ids.reshape(H, W); pts = pts.reshape(H, W, 3)
T = H // 8
idt = ids.reshape(T, 8, T, 8).transpose(0, 2, 1, 3).reshape(T * T, 64)
ptt = pts.reshape(T, 8, T, 8, 3).transpose(0, 2, 1, 3, 4).reshape(T * T, 64, 3)
wall = (idt >= 0) & (idt <= 4)
full = wall.all(1)
print("tiles", T * T, "all-wall tiles", full.sum())
P = ptt[full]
lo = P.min(1); hi = P.max(1); e = np.sort(hi - lo, axis=1)   # ascending
c = 0.5 * (lo + hi)
m = ctx.create_map(kind, n, 12345)
found, idx, d2, work = ctx.knn_profile(m, c)
rc = np.array([d2[i, :found[i]].max() for i in range(len(c))])
rad = np.sqrt(1.6 * rc)
def est(e1, e2): return 63 * (e1 + 2 * rad) * (e2 + 2 * rad) / (np.pi * rc)
whole = est(e[:, 2], e[:, 1]); half = est(e[:, 2] / 2, e[:, 1]); quarter = est(e[:, 2] / 4, e[:, 1]); q2 = est(e[:, 2] / 2, e[:, 1] / 2)
for name, v in (("whole", whole), ("half", half), ("quarter(long/4)", quarter), ("quarter(2x2)", q2)):
    print(f"{name:16s} est |S| p10 {np.percentile(v,10):.0f} p50 {np.percentile(v,50):.0f} p75 {np.percentile(v,75):.0f} p90 {np.percentile(v,90):.0f} p99 {np.percentile(v,99):.0f}  <=512: {(v<=512).mean()*100:.1f}%  <=1024: {(v<=1024).mean()*100:.1f}%")
al = e[:, 2] / np.sqrt(rc)
print("longest extent / r63: p10 %.2f p50 %.2f p90 %.2f p99 %.2f" % tuple(np.percentile(al, [10, 50, 90, 99])))
print("r63 p10 %.3f p50 %.3f p90 %.3f" % tuple(np.sqrt(np.percentile(rc, [10, 50, 90]))))
print("e_long p10 %.3f p50 %.3f p90 %.3f; e_mid p10 %.3f p50 %.3f p90 %.3f" % (*np.percentile(e[:, 2], [10, 50, 90]), *np.percentile(e[:, 1], [10, 50, 90])))
