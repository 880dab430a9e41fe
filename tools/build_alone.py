"""Diagnostics: the kernels of a map build with nothing else on the device (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000          # photons per map
for it in range(8):
    for kind in (0, 1):
        m = ctx.create_map(kind, n, 12345)
        torch.cuda.synchronize()
        m.destroy()
