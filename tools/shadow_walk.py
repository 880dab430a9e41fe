"""Statistics of the shadow pass of the 10 000 sphere scene: tests and grid cells per ray, wave-level rounds (PM_DEBUG print
of the counting variant), and the pass's time.  GPU only."""
import os, sys, time
os.environ["PM_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd.frame import GpuFrame

ctx = ra.Context(0)
ctx.upload_scene(ra.many_spheres_scene())
frame = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
for _ in range(2):
    frame.step()
torch.cuda.synchronize()
ctx.set_option("debug", 0)
ms = []
for _ in range(5):
    frame.build_maps(); st = frame.render(); ms.append((st.ms_primary, st.ms_secondary, st.ms_shadow, st.ms_total))
import hashlib
print("pane sha1", hashlib.sha1(frame.gather_pane().cpu().numpy().tobytes()).hexdigest())
print("ms primary/secondary/shadow/total:", [tuple(round(x, 3) for x in m) for m in ms])
for kind, m in enumerate(frame.maps):
    if m is not None:
        print("map", kind, "emit/build ms", m.times(), "photons sha1", hashlib.sha1(m.download().tobytes()).hexdigest(), len(m))
import ctypes
def trace_stats(label, reset=True):
    lib = ctx.lib
    if not hasattr(lib, "pm_trace_stats"):
        return
    out = (ctypes.c_ulonglong * 16)()
    lib.pm_trace_stats(out, 1 if reset else 0)
    names = ["serve calls", "lanes finished", "lanes fetched", "wave steps", "lane steps", "wave test rounds", "lane tests", "push turns", "loop turns"]
    print("trace stats", label, {n: out[i] for i, n in enumerate(names)})
trace_stats("(reset)")
frame.build_maps(); torch.cuda.synchronize(); trace_stats("both maps, one build")
for _ in range(3):
    torch.cuda.synchronize(); ta = time.perf_counter()
    frame.build_maps(); tt = [m.times() for m in frame.maps if m is not None]
    torch.cuda.synchronize(); print("maps alone: wall ms", round((time.perf_counter() - ta) * 1e3, 3), "emit/build per map", tt)
ms = []
for _ in range(5):
    st = frame.render(); ms.append((st.ms_primary, st.ms_secondary, st.ms_shadow, st.ms_total))
print("alone (maps kept) ms primary/secondary/shadow/total:", [tuple(round(x, 3) for x in m) for m in ms])
ctx.set_option("debug", 1)
frame.flags |= ra.PM_FLAG_COUNT_TESTS
st = frame.render()
print("tests primary", st.tests_primary_sphere, st.tests_primary_plane, "shadow", st.tests_shadow_sphere, st.tests_shadow_plane, "sp", st.absorb_points)
