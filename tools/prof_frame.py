"""Profiling driver: BASELINE config 2 frames (maps built once, frame rendered N times) for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
g = ctx.create_map(0, n, 12345); c = ctx.create_map(1, n, 12345)
rp = ctx.params(1440, 1440, 12345)
import ctypes
d_pane = ctypes.c_void_p()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc(ctypes.byref(d_pane), 1440 * 1440 * 4)
for i in range(frames):
    _, st = ctx.render(g, c, rp, d_pane=d_pane.value)
print({k: round(v, 3) if isinstance(v, float) else v for k, v in st.as_dict().items()})
