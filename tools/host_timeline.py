"""Where a step's wall time goes on the host: the calls of one bench step, timed one by one (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd.frame import GpuFrame
ctx = ra.Context(0)
ctx.upload_scene(ra.default_scene())
frame = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
for _ in range(5):
    frame.step()
torch.cuda.synchronize()
acc = {}
N = 50
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    for m in frame.maps:
        m.destroy()
    frame.maps = [None, None]
    t1 = time.perf_counter()
    frame.maps[1] = ctx.create_map(1, frame.n, frame.seed, frame.light_color, wait=False)
    t2 = time.perf_counter()
    frame.maps[0] = ctx.create_map(0, frame.n, frame.seed, frame.light_color, wait=False)
    t3 = time.perf_counter()
    st = frame.render()
    t4 = time.perf_counter()
    frame.gather_pane()
    tt = [m.times() for m in frame.maps]
    t5 = time.perf_counter()
    for k, v in (("destroy x2", t1 - t0), ("create caustic", t2 - t1), ("create global", t3 - t2), ("render (enqueue + wait)", t4 - t3), ("gather_pane + times", t5 - t4),
                 ("gpu ms_total", st.ms_total * 1e-3)):
        acc[k] = acc.get(k, 0.0) + v
torch.cuda.synchronize()
print(f"step wall {(time.perf_counter() - t_all) / N * 1e3:.3f} ms")
for k, v in acc.items():
    print(f"  {k:26s} {v / N * 1e3:.3f} ms")
