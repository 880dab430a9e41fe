"""Where the one-rank collective path of bench.py spends its time: per step, host wall of build_maps / render /
gather_pane with a device sync after each, with and without a process group.  usage: python tools/dist_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd.frame import GpuFrame

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
ctx = ra.Context(0)
ctx.upload_scene(ra.default_scene())


def run(tag, frame, steps=20):
    for _ in range(3):
        frame.step()
    torch.cuda.synchronize()
    t = [0.0, 0.0, 0.0]
    t_all = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); frame.build_maps(); torch.cuda.synchronize()
        b = time.perf_counter(); frame.render(); torch.cuda.synchronize()
        c = time.perf_counter(); frame.gather_pane(); torch.cuda.synchronize()
        d = time.perf_counter()
        t[0] += b - a; t[1] += c - b; t[2] += d - c
    torch.cuda.synchronize()
    print(f"{tag:28s} synced phases: maps {t[0] / steps * 1e3:.3f} render {t[1] / steps * 1e3:.3f} pane {t[2] / steps * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(steps):
        frame.step()
    torch.cuda.synchronize()
    print(f"{tag:28s} free-running step {(time.perf_counter() - t0) / steps * 1e3:.3f} ms")


f = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
run("no process group", f); f.close()
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
f = GpuFrame(ctx, 1440, 1440, 1000000, 12345)
run("group exists, not used", f); f.close()
for rep in (True, False):
    f = GpuFrame(ctx, 1440, 1440, 1000000, 12345, rank=0, world=1, dist=dist, replicate_maps=rep)
    run("dist, replicate" if rep else "dist, all-gather", f); f.close()
dist.destroy_process_group()
ctx.close()
