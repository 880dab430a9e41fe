"""Diagnostics: what one rank of N does per step on its own GPU when every rank emits and indexes both maps itself and traces
its share of the row blocks (no collectives: the lower bound of the N-GPU step, and its Amdahl split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
ctx = ra.Context(0); ctx.upload_scene(ra.default_scene())
# usage: rank_share.py [width height photons]   (default: BASELINE config 2; 7680 4320 10000000 = config 3's sizes)
W, H, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1440, 1440, 1000000)
seed = 12345
pane = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for world in (1, 2, 4, 8):
    ts = []
    for it in range(12 if n <= 2000000 else 6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = ctx.create_map(0, n, seed, wait=False); c = ctx.create_map(1, n, seed, wait=False)
        rp = ctx.params(W, H, seed); rp.rank = 0; rp.world = world
        _, st = ctx.render(g, c, rp, d_pane=pane.data_ptr())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, st.ms_primary + st.ms_secondary + st.ms_shadow, st.ms_gather))
        last = {k: round(getattr(st, k), 3) for k in ("ms_primary", "ms_secondary", "ms_shadow", "ms_gather_global", "ms_gather_caustic", "ms_group_global", "ms_group_caustic", "ms_resolve", "ms_total")}
        last["maps"] = [tuple(round(x, 3) for x in m.times()) for m in (g, c)]
        g.destroy(); c.destroy()
    ts = ts[4:] if n <= 2000000 else ts[2:]
    a = [sum(x[k] for x in ts) / len(ts) for k in range(3)]
    print("   ", last)
    print(f"world {world}: rank 0 step {a[0]:.3f} ms (passes before the gather {a[1]:.3f}, gather {a[2]:.3f})")
