"""Diagnostic (PM_WAVE_TIMES builds): per-wave busy time of the two gather kernels, read from PM_WAVE_DUMP."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(2, 8192, 4)
for kind, name in enumerate(("global", "caustic")):
    t0, t1, nq, xcd = (a[kind, :, i].astype(np.int64) for i in range(4))
    use = nq >= 0
    dur = (t1 - t0) & 0xFFFFFFFF
    start = t0.min(); end = ((t1 - start) & 0xFFFFFFFF)
    span = end.max()
    print(f"{name}: span {span} ticks; wave busy mean {dur.mean():.0f} p50 {np.median(dur):.0f} p90 {np.percentile(dur,90):.0f} max {dur.max()}; "
          f"queries/wave mean {nq.mean():.1f} min {nq.min()} max {nq.max()}; late start max {((t0-start)&0xFFFFFFFF).max()}")
    for x in range(8):
        m = xcd == x
        print(f"   xcd {x}: busy mean {dur[m].mean():.0f} max {dur[m].max()} end max {end[m].max()} queries {nq[m].sum()}")
