#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE config 2: the reference's default Cornell scene at
1440x1440, 1 spp, 1,000,000 global + 1,000,000 caustic photons, k = 63, 25 shadow rays, depth <= 3.

A step is ONE WHOLE FRAME of renderScene (reference src/Raytrace.cpp:301-343): emit both photon maps,
build their indices, trace / shade every pixel, pane complete on rank 0.  The scene is resident in HBM
before the timed region; nothing else is.  With --gpus N the one frame is split over N ranks (strong
scaling): emission attempts by range + one all-gather of 32-byte photon records, image rows by
interleaved 16-row blocks + a gather of the row blocks to rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints one JSON line (rank 0).  `value` = Mrays/s = (closest-hit + shadow any-hit queries of the whole
frame) / step time, all ranks.  `roofline` is for the dominant kernel (the k-NN gather), timed with HIP
events on the library's stream inside the same timed steps.  `cpu_baseline` is the CPU oracle
(oracle/pm_oracle.c, a restatement of the reference pinned against the compiled reference) rendering a
bounded row sample of the SAME frame with the SAME photons on the host cores -- the checker used as a
yardstick, never as the product path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_GATHER = 63 * 32      # k photon records of 32 B
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
WORKLOAD = {"default": "default Cornell scene", "spheres10k": "five walls + 10,000 diffuse spheres (BASELINE config 5 scene),",
            "caustics": "five walls + five glass spheres (BASELINE config 4 scene),"}


def latest_pmc_summary():
    """profiles/rNN/pmc_per_dispatch_avg.json of the latest round that has one."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_per_dispatch_avg.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r01", "pmc_per_dispatch_avg.json")


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes of this same command (tools/profile_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs; rocprofv3 reports KB; gfx950 correction: FETCH_SIZE tallies the
    128-byte read requests at 64 bytes, so it is doubled -- MI355X_MICROARCH.md, HBM section)."""
    path = latest_pmc_summary()
    try:
        with open(path) as f:
            pmc = json.load(f)
        for name, c in pmc.items():
            if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, f"2 x FETCH_SIZE + WRITE_SIZE of {os.path.relpath(path, ROOT)} (not collected live)"
    except (OSError, ValueError):
        pass
    return None, "no PMC summary found"


# One SIMD issues an FP64 wave-instruction every ~2.2 ns when all of them are busy (profiles/r01/valu_rates.txt): 1024 SIMDs.
VALU_PEAK_GINSTR_S = 1024 / 2.2


def pmc_valu_issue(kernel, launch_ms):
    """What actually bounds the gather kernels: vector instructions issued per second, from the
    committed SQ_INSTS_VALU count of one launch of `kernel` over the launch time measured live, against the measured
    issue rate of the whole device.  The other map's kernel shares the device during that time."""
    path = latest_pmc_summary()
    try:
        with open(path) as f:
            pmc = json.load(f)
        for name, c in pmc.items():
            if kernel in name and "SQ_INSTS_VALU" in c and launch_ms > 0:
                ach = c["SQ_INSTS_VALU"] / (launch_ms * 1e-3) / 1e9
                return {"achieved": ach, "peak": VALU_PEAK_GINSTR_S, "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINSTR_S,
                        "note": f"SQ_INSTS_VALU per launch from {os.path.relpath(path, ROOT)} (default scene; not collected live) "
                                "/ live launch time; peak = 1024 SIMDs x one FP64 instruction per 2.2 ns (profiles/r01/valu_rates.txt); "
                                "the two maps' kernels run side by side, each taking its share of that peak"}
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(frame, args, cores, scene):
    """Oracle ("port") on the host cores: every `stride`-th row of the same frame, same photon maps."""
    import numpy as np
    from oracle import orc_py
    gmap = orc_py.Map(frame.maps[0].download())
    cmap = orc_py.Map(frame.maps[1].download())
    rp = orc_py.params(args.width, args.height, args.seed)
    stride = args.cpu_row_stride
    # start inside the box (the top/bottom quarter of a 1440x1440 frame misses the scene)
    t0 = time.perf_counter()
    pane, st = orc_py.render(scene, gmap, cmap, rp, y0=stride // 2, y1=args.height, ystep=stride, nthreads=cores)
    dt = time.perf_counter() - t0
    rows = len(range(stride // 2, args.height, stride))
    rays = st.closest_hit_queries + st.shadow_queries
    gmap.close(); cmap.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"rows {stride // 2}::{stride} of the same {args.width}x{args.height} frame ({rows} rows, {rays} rays, "
                      f"{st.gathers} gathers) with the same 2x{args.photons} photons (CPU index build not timed), {dt:.2f} s",
            "gathers_per_s": st.gathers / dt, "seconds": dt}, pane


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--maps", default="auto", choices=["auto", "allgather", "replicate"],
                    help="with more than one rank: emission split by attempt range + one all-gather of the shards (the path as specified), "
                         "or every rank emits every map itself and nothing is exchanged (same maps, bit for bit); auto times a few "
                         "untimed steps of each before the warmup and keeps the faster (all ranks agree on the slowest rank's times)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1440)
    ap.add_argument("--height", type=int, default=1440)
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--scene", default="default", choices=["default", "spheres10k", "caustics"],
                    help="default = BASELINE config 2 (the bench line); spheres10k / caustics = the synthetic scenes of configs 5 / 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="compare the CPU sample rows with the GPU pane")
    ap.add_argument("--dump-pane", default="", help="rank 0 writes the final pane (uint32 raw) to this file")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: whatever libraries write to fd 1 meanwhile (RCCL prints a version
    # banner there when the process group starts) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd.frame import GpuFrame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # PM_FORCE_DIST=1 (tests): take the collective path even with one rank, to exercise RCCL on device buffers
    use_dist = world > 1 or os.environ.get("PM_FORCE_DIST") == "1"
    # The library's context first: a process has 4 hardware queues and streams are dealt onto them in the order they are
    # created.  Created after RCCL's, the library's three streams end up sharing queues and the gathers that are meant to
    # run side by side run one behind the other (measured with one rank: 3.8 - 4.1 ms per step instead of 3.0).
    ctx = ra.Context(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    scene = {"default": ra.default_scene, "spheres10k": ra.many_spheres_scene, "caustics": ra.caustics_scene}[args.scene]()
    ctx.upload_scene(scene)                   # the input that is resident before the timed region
    frame = GpuFrame(ctx, args.width, args.height, args.photons, args.seed, rank=rank, world=world,
                     dist=dist if use_dist else None, replicate_maps=args.maps == "replicate")
    maps_mode, maps_tune = args.maps, None

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist and args.maps == "auto":
        # which way of making the maps is faster HERE (ranks, links, photon count): a few untimed steps of each
        maps_tune = {}
        for mode in ("allgather", "replicate"):
            frame.replicate_maps = mode == "replicate"
            for _ in range(2):
                frame.step()
            sync()
            ta = time.perf_counter()
            for _ in range(5):
                frame.step()
            sync()
            tm = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            maps_tune[mode] = float(tm[0]) / 5 * 1e3
        maps_mode = min(maps_tune, key=maps_tune.get)
        frame.replicate_maps = maps_mode == "replicate"
    elif not use_dist:
        maps_mode = "local"
    for _ in range(args.warmup):
        frame.step()
    sync()
    phase = {k: 0.0 for k in ("ms_primary", "ms_secondary", "ms_shadow", "ms_gather_global", "ms_gather_caustic", "ms_gather", "ms_group_global",
                              "ms_group_caustic", "ms_resolve", "ms_total")}
    t_maps = 0.0
    map_ms = [[0.0, 0.0], [0.0, 0.0]]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        frame.build_maps()      # emission + build are left on the library's build streams; render waits for them where it needs them
        t_maps += time.perf_counter() - ta
        st = frame.render()
        frame.gather_pane()
        for k in phase:
            phase[k] += getattr(st, k)
        for kind, m in enumerate(frame.maps):      # device times of the two phases of each map's build (HIP events on its build stream)
            if m is not None:
                te, tb = m.times()
                map_ms[kind][0] += te; map_ms[kind][1] += tb
    sync()
    dt = time.perf_counter() - t0

    # whole-job totals
    tot = torch.tensor([dt, float(st.closest_hit_queries), float(st.shadow_queries), float(st.gathers), float(st.absorb_points),
                        phase["ms_gather_global"], phase["ms_gather_caustic"], phase["ms_total"], t_maps], dtype=torch.float64, device="cuda")
    if use_dist:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0]); rays = float(sm[1] + sm[2]); gathers = float(sm[3]); sp_local = float(tot[4])
        g_ms, c_ms = float(mx[5]), float(mx[6])
    else:
        rays = float(tot[1] + tot[2]); gathers = float(tot[3]); sp_local = float(tot[4]); g_ms, c_ms = float(tot[5]), float(tot[6])

    out = None
    if rank == 0:
        step_s = dt / args.steps
        # dominant kernel: the group gather k_gather_tile<map> (one launch per map and step; the two maps run side by side
        # on two streams).  Units per launch = the shade points that launch answered on this rank.
        k = 0 if phase["ms_group_global"] >= phase["ms_group_caustic"] else 1
        dom = ("k_gather_tile<0> (global map)", "k_gather_tile<1> (caustic map)")[k]
        dom_ms = (phase["ms_group_global"], phase["ms_group_caustic"])[k] / args.steps
        dom_units = (st.grouped_global, st.grouped_caustic)[k]
        achieved = ALGO_BYTES_PER_GATHER * dom_units / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic(("k_gather_tile<0>", "k_gather_tile<1>")[k])
        out = {
            "metric": "Mrays/sec + photon-gathers/sec at 1440p, 1/2/4/8 GPUs; frame time (s)",
            "value": rays / step_s / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD[args.scene]} {args.width}x{args.height}, 1 spp, {args.photons} global + {args.photons} caustic photons, "
                                   f"k=63, 25 shadow rays, depth<=3{' (BASELINE config 2)' if args.scene == 'default' else ''}; step = emit both maps + build indices + trace/shade + pane on rank 0",
                       "seed": args.seed, "parallelism": f"rows x{world} (16-row blocks), " + (f"emission attempts x{world} + all-gather" if maps_mode == "allgather" else
                                                                                   "every map emitted on every rank, no exchange" if maps_mode == "replicate" else "one rank"),
                       "maps_autotune_ms_per_step": maps_tune},
            "gathers_per_s": gathers / step_s, "frame_time_s": step_s, "rays_per_frame": rays, "gathers_per_frame": gathers,
            "phases_ms_rank0": {**{k: v / args.steps for k, v in phase.items()}, "maps_enqueue_host_wall": t_maps / args.steps * 1e3,
                                "map_global_emit": map_ms[0][0] / args.steps, "map_global_index": map_ms[0][1] / args.steps,
                                "map_caustic_emit": map_ms[1][0] / args.steps, "map_caustic_index": map_ms[1][1] / args.steps},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "note": f"algorithmic bytes = {ALGO_BYTES_PER_GATHER} B x {int(dom_units)} gathers answered per launch; avg launch {dom_ms:.3f} ms "
                                 f"(HIP events on the stream the kernel is launched on; the other map's gather runs beside it on a second stream)",
                         "valu_issue": pmc_valu_issue(("k_gather_tile<0>", "k_gather_tile<1>")[k], dom_ms)
                         if (args.scene, args.width, args.height, args.photons) == ("default", 1440, 1440, 1000000) else None},
        }
        if not args.no_cpu_baseline and world == 1:
            cores = args.cpu_cores or min(os.cpu_count() or 1, 16)
            cb, cpane = cpu_baseline(frame, args, cores, scene)
            if args.check:
                gp = frame.pane.cpu().numpy().view(np.uint32)
                rows = slice(args.cpu_row_stride // 2, args.height, args.cpu_row_stride)
                cb["sample_rows_bit_identical_to_gpu"] = bool(np.array_equal(gp[rows], cpane[rows]))
            out["cpu_baseline"] = cb
        if args.dump_pane:
            frame.step()[1].cpu().numpy().tofile(args.dump_pane)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    frame.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
