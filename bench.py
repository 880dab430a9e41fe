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


# One SIMD issues an FP64 wave-instruction every ~2.2 ns when all of them are busy (tools/microbench/valu_rates.hip, re-run in round 3:
# profiles/r03/valu_rates.txt): 1024 SIMDs.  The spec-derived figure beside it: 78.6 TFLOP/s FP64 vector / 2 (an FMA counts twice) / 64 lanes.
VALU_PEAK_GINSTR_S = 1024 / 2.2
VALU_SPEC_GINSTR_S = 78.6e12 / 2 / 64 / 1e9
N_SIMDS = 1024
# FP64 vector peak 78.6 TFLOP/s counts an FMA as two; contraction is off here (bit-exactness), so 39.3 T non-fused flop/s
FP64_PEAK_TFLOPS = 39.3
FLOP_PER_SPHERE_TEST, FLOP_PER_PLANE_TEST = 22, 6
PMC_PASSES = (("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"), ("FETCH_SIZE",), ("WRITE_SIZE",))   # FETCH_SIZE and WRITE_SIZE do not fit one pass


def under_profiler():
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))


def live_pmc(args):
    """Counters of THIS build on THIS box, collected now: one rocprofv3 --pmc pass per counter set (as MI355X_MICROARCH.md
    prescribes) over a short run of this same script and workload, as child processes.  Returns ({kernel: {counter: average
    per launch}}, note); ({}, why) when the profiler is not there or a pass fails -- the line then carries nulls, never a
    number from an earlier build."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return {}, "rocprofv3 not found"
    if under_profiler():
        return {}, "already running under a profiler"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--pmc-child", "--scene", args.scene,
             "--width", str(args.width), "--height", str(args.height), "--photons", str(args.photons), "--seed", str(args.seed)]
    acc = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for n, counters in enumerate(PMC_PASSES):
        out = tempfile.mkdtemp(prefix="pm_pmc_", dir="/tmp")
        try:
            r = subprocess.run([tool, "--output-format", "csv", "--pmc", *counters, "-d", out, "-o", "pmc", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {}, f"rocprofv3 pass {n + 1} ({' '.join(counters)}) failed: {(r.stderr or r.stdout)[-200:]}"
            per = {}
            for path in files:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        key = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Dispatch_Id"], row["Counter_Name"])
                        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            lists = {}
            for (k, _, c), v in per.items():
                lists.setdefault((k, c), []).append(v)
            for (k, c), v in lists.items():
                acc.setdefault(k, {})[c] = sum(v) / len(v)
                acc[k]["launches_" + c] = len(v)
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
            return {}, f"rocprofv3 pass {n + 1} failed: {e!r}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return acc, ("collected in this run: rocprofv3 --pmc, one pass each for " + " | ".join(" ".join(c) for c in PMC_PASSES) +
                 ", over 3 steps of this workload in a child process; averages per launch")


def kernel_counters(pmc, name):
    """Counters of kernel `name` ("k_gather_tile<0>", "k_primary"): the exact kernel, not one whose name merely begins alike."""
    for k, c in pmc.items():
        if k == name or k == "pm::" + name or (("<" not in name) and k.split("<")[0] in (name, "pm::" + name)):
            return c
    return {}


def hbm_traffic(c):
    """HBM-side bytes per launch: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE tallies the 128-byte
    read requests at 64 bytes and is doubled (MI355X_MICROARCH.md, HBM section)."""
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
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
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes (roofline.traffic and the instruction counts are then null)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) a child run under rocprofv3: the timed steps only, no counting frame, no host-pane loop")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the CPU baseline (default: the cores this process may run on)")
    ap.add_argument("--check", action="store_true", help="compare the CPU sample rows with the GPU pane")
    ap.add_argument("--check-hash", action="store_true", help="add the SHA-1 of the last timed step's pane to the line (A/B runs: same bits?)")
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
    last_pane = None
    map_ms = [[0.0, 0.0], [0.0, 0.0]]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        frame.build_maps()      # emission + build are left on the library's build streams; render waits for them where it needs them
        t_maps += time.perf_counter() - ta
        st = frame.render()
        last_pane = frame.gather_pane()
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

    # after the timed region, one more frame that counts the primitive tests of the primary and the shadow pass (same pane,
    # counting kernels: the intersection roofline needs tests, not rays, once a scene is walked through the sphere grid)
    tests = None
    if not use_dist and not args.pmc_child:
        frame.flags |= ra.PM_FLAG_COUNT_TESTS
        stc = frame.render()
        frame.flags &= ~ra.PM_FLAG_COUNT_TESTS
        tests = {"primary": (stc.tests_primary_sphere, stc.tests_primary_plane), "shadow": (stc.tests_shadow_sphere, stc.tests_shadow_plane)}

    out = None
    if rank == 0:
        step_s = dt / args.steps
        ph = {k: v / args.steps for k, v in phase.items()}
        pmc, pmc_note = ({}, "skipped (--no-pmc)") if args.no_pmc or use_dist else live_pmc(args)
        # Every kernel class of the step with a device time of its own (HIP events on the stream it runs on), its roofline, and
        # the dominant one: the class with the most device time per step.  The two maps' launches of one kernel template are
        # one class (two launches per step): which of the two is the longer flips from run to run with how they overlap.
        # (the emission kernel depends on the scene: through a sphere grid k_emit_stream, else k_emit_run; k_emit with PM_EMIT_RUN=0)
        em_name = ("k_emit_stream" if kernel_counters(pmc, "k_emit_stream") or args.scene == "spheres10k" else
                   "k_emit" if (pmc and kernel_counters(pmc, "k_emit") and not kernel_counters(pmc, "k_emit_run")) else "k_emit_run")
        pq_name = "k_gather" if (pmc and not kernel_counters(pmc, "k_gather_pair<0>")) else "k_gather_pair"
        n_maps = 2
        left = [st.left_global, st.left_caustic]
        classes = {
            "k_gather_tile": {"launches": n_maps, "ms": (ph["ms_group_global"] + ph["ms_group_caustic"]) / n_maps, "pmc": ("k_gather_tile<0>", "k_gather_tile<1>"),
                              "gathers": (st.grouped_global + st.grouped_caustic) / n_maps,
                              "what": "group gather (pm_tile.h), one launch per map, side by side on two streams"},
            pq_name: {"launches": n_maps, "ms": max(0.0, (ph["ms_gather_global"] - ph["ms_group_global"]) + (ph["ms_gather_caustic"] - ph["ms_group_caustic"])) / n_maps,
                      "pmc": (pq_name + "<0>", pq_name + "<1>"), "gathers": sum(left) / n_maps,
                      "what": "per-query gather of what the group kernel left, behind it on its stream (the launch beside the group kernels, "
                              f"{st.early_points} bounced pixels' shade points per map, has no events of its own; the counters average both launches)"},
            "k_shadow_rays": {"launches": 1, "ms": ph["ms_shadow"], "pmc": ("k_shadow_rays",), "tests": tests["shadow"] if tests else None, "what": "shadow any-hit pass, one ray per lane"},
            "k_primary": {"launches": 1, "ms": ph["ms_primary"], "pmc": ("k_primary",), "tests": tests["primary"] if tests else None, "what": "primary closest-hit pass"},
            em_name: {"launches": n_maps, "ms": (map_ms[0][0] + map_ms[1][0]) / args.steps / n_maps, "pmc": (em_name,),
                      "what": "photon emission of one map (events around the emission kernel + k_scan_blocks + k_emit_scatter: the first is nearly all of it)"},
        }
        clock_hz = torch.cuda.get_device_properties(local_rank).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(local_rank), "clock_rate") else 2.4e9

        def class_counters(names):
            got = [kernel_counters(pmc, n) for n in names]
            got = [g for g in got if g]
            keys = set().union(*[g.keys() for g in got]) if got else set()
            return {k: sum(g.get(k, 0.0) for g in got) / len(got) for k in keys}

        def roofline_of(name, k):
            c = class_counters(k["pmc"])
            ms = k["ms"]
            valu = c.get("SQ_INSTS_VALU")
            issue = None
            if valu and ms > 0:
                ach = valu / (ms * 1e-3) / 1e9
                issue = {"achieved": ach, "peak": VALU_PEAK_GINSTR_S, "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINSTR_S,
                         "frac_of_spec_fp64_rate": ach / VALU_SPEC_GINSTR_S, "spec_peak": VALU_SPEC_GINSTR_S,
                         "valu_instr_per_launch": valu, "salu_instr_per_launch": c.get("SQ_INSTS_SALU")}
            occ = None
            if c.get("SQ_WAVE_CYCLES") and ms > 0:
                # SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md): x 4 / (launch time x clock x SIMDs)
                occ = {"waves_per_simd_achieved": 4.0 * c["SQ_WAVE_CYCLES"] / (ms * 1e-3 * clock_hz * N_SIMDS), "waves_per_launch": c.get("SQ_WAVES"),
                       "sq_wave_cycles": c["SQ_WAVE_CYCLES"], "sq_busy_cycles": c.get("SQ_BUSY_CYCLES"), "clock_hz_assumed": clock_hz}
            r = {"kernel": name, "launches_per_step": k["launches"], "ms_per_launch": ms, "ms_per_step": ms * k["launches"], "what": k["what"],
                 "traffic": hbm_traffic(c), "valu_issue": issue, "occupancy": occ}
            if "gathers" in k:
                hbm_ach = ALGO_BYTES_PER_GATHER * k["gathers"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                hbm = {"achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS,
                       "note": f"SURVEY 8d's figure: {ALGO_BYTES_PER_GATHER} B x {k['gathers']:.0f} gathers answered per launch / launch time; the maps (64 MB + tables "
                               "each at 1M photons) stay in L2 / Infinity Cache, so HBM is not what binds this kernel"}
                # the kernel is bound by its waves' instruction streams (FP64 selection and radiance arithmetic, memory latency): the issue
                # fraction is the one reported; without counters (no rocprofv3, a failed pass) the HBM figure stands in, named as such
                if issue:
                    r.update({"bound": "valu", "achieved": issue["achieved"], "peak": VALU_PEAK_GINSTR_S, "unit": "G wave-instr/s", "frac": issue["frac"], "hbm": hbm})
                else:
                    r.update({"bound": "hbm", "achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS, "hbm": hbm,
                              "fallback": "no instruction counters in this run (" + pmc_note + "): SURVEY 8d's algorithmic HBM fraction instead of the issue fraction"})
            elif k.get("tests") is not None or name in ("k_shadow_rays", "k_primary"):
                sph, pl = k.get("tests") or (0, 0)
                flop = sph * FLOP_PER_SPHERE_TEST + pl * FLOP_PER_PLANE_TEST
                ach = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                r.update({"bound": "valu", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "sphere_tests": sph, "plane_tests": pl,
                          "note": f"{sph} sphere tests x {FLOP_PER_SPHERE_TEST} + {pl} plane tests x {FLOP_PER_PLANE_TEST} flop (counted on the device by the counting variant "
                                  "of the pass, same frame) / launch time; peak = FP64 vector without FMA contraction"})
            else:
                r.update({"bound": "valu", "achieved": issue["achieved"] if issue else None, "peak": VALU_PEAK_GINSTR_S, "unit": "G wave-instr/s",
                          "frac": issue["frac"] if issue else None})
            return r

        roof_all = [roofline_of(n, k) for n, k in classes.items()]
        roof = dict(max(roof_all, key=lambda r: r["ms_per_step"]))
        roof["traffic_note"] = pmc_note
        roof["note"] = (f"dominant kernel class of the step by device time ({roof['ms_per_step']:.3f} ms per step in {roof['launches_per_step']} launch(es), HIP events on the "
                        "stream each launch runs on; kernels of other streams share the device during that time); valu peak = 1024 SIMDs x one FP64 "
                        f"wave-instruction per 2.2 ns measured (profiles/r03/valu_rates.txt), spec-derived {VALU_SPEC_GINSTR_S:.0f} G/s beside it; a kernel whose mix is "
                        "mostly 32-bit integer and move instructions (1.1 ns) can exceed the FP64 figure")
        # the intersection passes against the FP64 peak, whichever kernel dominates (north_star: VALU / occupancy for intersection)
        inter = {r["kernel"]: {"ms": r["ms_per_launch"], "sphere_tests": r.get("sphere_tests"), "plane_tests": r.get("plane_tests"), "tflops": r["achieved"],
                               "frac_of_fp64_peak": r["frac"], "valu_issue_frac": r["valu_issue"]["frac"] if r["valu_issue"] else None,
                               "valu_instr_per_launch": r["valu_issue"]["valu_instr_per_launch"] if r["valu_issue"] else None,
                               "waves_per_simd_achieved": r["occupancy"]["waves_per_simd_achieved"] if r["occupancy"] else None,
                               "waves_per_launch": r["occupancy"]["waves_per_launch"] if r["occupancy"] else None}
                 for r in roof_all if r["kernel"] in ("k_primary", "k_shadow_rays")} if tests else None
        per_query = None
        pqc = class_counters(classes[pq_name]["pmc"])
        if pqc.get("SQ_INSTS_VALU"):
            n_pq = [st.absorb_points - st.grouped_global, st.absorb_points - st.grouped_caustic]
            # per map and step two launches of the kernel: beside the group kernel (the points never grouped) and behind it (what it left);
            # the counters are the average over both launches of both maps, so x 4 is the step's total
            tot = 4.0 * pqc["SQ_INSTS_VALU"]
            per_query = {"gathers_per_step": int(sum(n_pq)), "valu_instr_per_gather": tot / max(1, sum(n_pq)), "kernel": pq_name + "<0/1>",
                         "left_by_group_kernels": left, "never_grouped_per_map": st.early_points,
                         "note": "SQ_INSTS_VALU averaged over the kernel's launches (live PMC pass) x 4 launches per step / shade points the group kernels did not answer"}
        config_tag = " (BASELINE config 2)" if (args.scene, args.width, args.height, args.photons) == ("default", 1440, 1440, 1000000) else ""
        out = {
            "metric": "Mrays/sec + photon-gathers/sec at 1440p, 1/2/4/8 GPUs; frame time (s)",
            "value": rays / step_s / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD[args.scene]} {args.width}x{args.height}, 1 spp, {args.photons} global + {args.photons} caustic photons, "
                                   f"k=63, 25 shadow rays, depth<=3{config_tag}; step = emit both maps + build indices + trace/shade + pane on rank 0",
                       "seed": args.seed, "parallelism": f"rows x{world} (16-row blocks), " + (f"emission attempts x{world} + all-gather" if maps_mode == "allgather" else
                                                                                   "every map emitted on every rank, no exchange" if maps_mode == "replicate" else "one rank"),
                       "maps_autotune_ms_per_step": maps_tune},
            "gathers_per_s": gathers / step_s, "frame_time_s": step_s, "rays_per_frame": rays, "gathers_per_frame": gathers,
            "phases_ms_rank0": {**ph, "maps_enqueue_host_wall": t_maps / args.steps * 1e3,
                                "map_global_emit": map_ms[0][0] / args.steps, "map_global_index": map_ms[0][1] / args.steps,
                                "map_caustic_emit": map_ms[1][0] / args.steps, "map_caustic_index": map_ms[1][1] / args.steps},
            "roofline": roof, "roofline_all": roof_all, "intersection": inter, "per_query_gather": per_query,
        }
        if not use_dist and not args.pmc_child:
            # the reference's boundary hands back a host pane (uint32 *pane of renderScene): the same step with the copy
            import numpy as _np
            host = _np.empty((args.height, args.width), dtype=_np.uint32)
            rp = ctx.params(args.width, args.height, args.seed)
            th = 0.0
            for it in range(6):
                if it == 1:
                    th = time.perf_counter()           # (the first one allocates the device-side pane)
                frame.build_maps()
                ctx.render(frame.maps[0], frame.maps[1], rp, pane=host)
            out["ms_per_step_host_pane"] = (time.perf_counter() - th) / 5 * 1e3
        if not args.no_cpu_baseline and world == 1:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = args.cpu_cores or max(1, min(avail, int(os.environ.get("OMP_NUM_THREADS", avail))))
            cb, cpane = cpu_baseline(frame, args, cores, scene)
            cb["host_cpus_visible"] = os.cpu_count()
            if args.check:
                gp = frame.pane.cpu().numpy().view(np.uint32)
                rows = slice(args.cpu_row_stride // 2, args.height, args.cpu_row_stride)
                cb["sample_rows_bit_identical_to_gpu"] = bool(np.array_equal(gp[rows], cpane[rows]))
            out["cpu_baseline"] = cb
        if args.check_hash and last_pane is not None:
            import hashlib
            out["pane_sha1"] = hashlib.sha1(last_pane.cpu().numpy().tobytes()).hexdigest()
        if args.dump_pane and last_pane is not None:
            last_pane.cpu().numpy().tofile(args.dump_pane)      # the pane the last timed step gathered on rank 0
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    frame.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
