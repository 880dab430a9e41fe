"""Diagnostics: how long the two map builds of a frame take on their own (nothing else on the device), and the frame after them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
import raytracer_amd as ra
hip = ctypes.CDLL("libamdhip64.so")
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
pane = torch.zeros((1440, 1440), dtype=torch.int32, device="cuda")
rp = ctx.params(1440, 1440, 12345)
for mode in ("both maps, then wait, then frame", "one map at a time, waits between"):
    ts = []
    for it in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode.startswith("both"):
            g = ctx.create_map(0, 1000000, 12345, wait=False); c = ctx.create_map(1, 1000000, 12345, wait=False)
            t1 = time.perf_counter()
            hip.hipDeviceSynchronize(); t2 = time.perf_counter(); t3 = t2
        else:
            g = ctx.create_map(0, 1000000, 12345, wait=False); t1 = time.perf_counter(); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
            c = ctx.create_map(1, 1000000, 12345, wait=False); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
        _, st = ctx.render(g, c, rp, d_pane=pane.data_ptr())
        torch.cuda.synchronize(); t4 = time.perf_counter()
        g.destroy(); c.destroy()
        ts.append((t1 - t0, t2 - t0, t3 - t0, t4 - t3, st.ms_total))
    a = np.array(ts[3:]).mean(0) * 1e3
    print(f"{mode}: enqueue {a[0]:.3f} ms, first wait over at {a[1]:.3f} ms, builds done at {a[2]:.3f} ms, frame after that {a[3]:.3f} ms (device {a[4]/1e3:.3f})")
