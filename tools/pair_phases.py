"""Diagnostics (library built with -DPM_PAIR_STAMPS, PM_LIB pointing at it): where a wave of the two-queries-per-wave gather
spends its clock ticks, by phase, over one frame of the default bench workload."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd.frame import GpuFrame
scene = {"default": ra.default_scene, "spheres10k": ra.many_spheres_scene, "caustics": ra.caustics_scene}[sys.argv[1] if len(sys.argv) > 1 else "default"]()
ctx = ra.Context(0); ctx.upload_scene(scene)
frame = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
for _ in range(3):
    frame.step()
out = (ctypes.c_ulonglong * 8)()
ctx.lib.pm_pair_ticks(out, 1)
frame.step()
ctx.lib.pm_pair_ticks(out, 0)
names = ["level + bounds", "block cells + list", "scan", "select", "radiance sum", "walk (fallback, whole wave)"]
tot = sum(out[i] for i in range(6))
for i, n in enumerate(names):
    print(f"{n:32s} {out[i] / 1e6:10.1f} M ticks {100.0 * out[i] / tot:5.1f} %")
print("pairs", out[7], "fallback walks", out[6], "ticks per pair (fast path)", round(sum(out[i] for i in range(5)) / max(1, out[7])), "ticks per walk", round(out[5] / max(1, out[6])))
