"""Randomised parity sweep on the GPU box: random frame sizes, photon counts, seeds, cameras, light colours and flags on
the three scenes, every pane compared bit for bit with the oracle's.  usage: fuzz_parity.py [cases] [seed] [large]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from oracle import orc_py as orc

def sweep(cases, seed, verbose=True, large=False):
    """Runs `cases` random configurations; returns the list of mismatch descriptions (empty = parity)."""
    rng = np.random.default_rng(seed)
    scenes = {"default": ra.default_scene(), "caustics": ra.caustics_scene(), "spheres300": ra.many_spheres_scene(300, 5)}
    ctxs = {}
    for name, sc in scenes.items():
        ctxs[name] = ra.Context(0)
        ctxs[name].upload_scene(sc)
    bad = []
    t0 = time.time()
    try:
        for k in range(cases):
            name = list(scenes)[int(rng.integers(0, 3))]
            ctx, scene = ctxs[name], scenes[name]
            if large:   # big octrees, mostly grouped gathers
                w, h = int(rng.integers(300, 1000)), int(rng.integers(300, 700))
                n = int(rng.choice([100000, 300000, 1000000]))
            else:
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
                n = int(rng.choice([0, 5, 63, 64, 500, 3000, 20000, 60000]))
            sd = int(rng.integers(0, 2 ** 40))
            flags = int(rng.choice([0, 0, 0, 1, 2, 4]))
            cam = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), float(rng.uniform(-15, -6)))
            lc = tuple(float(x) for x in rng.uniform(0.1, 1.0, 3))
            g, c = ctx.create_map(0, n, sd, lc, wait=False), ctx.create_map(1, n, sd, lc, wait=False)
            pane, st = ctx.render(None if flags & 1 else g, None if flags & 2 else c, ctx.params(w, h, sd, cam=cam, light_color=lc, flags=flags))
            pg, pc = g.download(), c.download()
            ok_maps = all(np.array_equal(got.view(np.uint8), np.asarray(orc.emit(scene, kind, sd, n, light_color=lc)[0]).view(np.uint8))
                          for kind, got in ((0, pg), (1, pc)))
            og, oc = orc.Map(pg), orc.Map(pc)
            want, ost = orc.render(scene, og, oc, orc.params(w, h, sd, cam=cam, light_color=lc, flags=flags), nthreads=16)
            ok = ok_maps and np.array_equal(pane, want) and (st.closest_hit_queries, st.shadow_queries, st.gathers) == (
                ost.closest_hit_queries, ost.shadow_queries, ost.gathers)
            if not ok:
                bad.append(f"case {k}: scene {name} {w}x{h} n={n} seed={sd} flags={flags} cam={cam} lc={lc} maps_ok={ok_maps} "
                           f"differing px={int((pane != want).sum())}")
                if verbose:
                    print("MISMATCH", bad[-1], flush=True)
            g.destroy(); c.destroy(); og.close(); oc.close()
            if verbose and k % 20 == 19:
                print(f"{k + 1} cases, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
    finally:
        for ctx in ctxs.values():
            ctx.close()
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    found = sweep(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1, large=len(sys.argv) > 3 and sys.argv[3] == "large")
    print(f"done: {n_cases} cases, {len(found)} mismatches")
    sys.exit(1 if found else 0)
