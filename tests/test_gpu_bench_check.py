"""BASELINE.json's sizes in the suite: bench.py --check renders the 1440 x 1440 frame with 1M + 1M photons on the GPU and
a sample of its rows with the oracle on the same photons, and says whether the rows are bit-identical.  One case per scene
of configs 2, 4 and 5 (the oracle scans 10 005 primitives per ray on the last: fewer rows)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scene,stride", [("default", 4), ("caustics", 4), ("spheres10k", 48)])
def test_full_size_frame_rows_equal_the_oracle(scene, stride):
    # the default scene also goes through the live counter passes (rocprofv3 child processes); the others skip them to save time
    extra = [] if scene == "default" else ["--no-pmc"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--scene", scene, "--check",
                        "--cpu-row-stride", str(stride)] + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cb = j["cpu_baseline"]
    assert cb["sample_rows_bit_identical_to_gpu"] is True, cb
    assert f"{len(range(stride // 2, 1440, stride))} rows" in cb["sample"]
    assert j["rays_per_frame"] > 10e6 and j["gathers_per_frame"] > 1e6 and j["value"] > 0
    assert "1440x1440" in j["config"]["workload"] and "1000000 global" in j["config"]["workload"]
    roof = j["roofline"]
    classes = {r["kernel"]: r for r in j["roofline_all"]}
    # every kernel class of the step with a device time of its own is a candidate: gathers, passes, emission
    assert {"k_gather_tile", "k_gather_pair", "k_shadow_rays", "k_primary"} <= set(classes) and any(k.startswith("k_emit") for k in classes)
    assert roof["kernel"] == max(classes.values(), key=lambda r: r["ms_per_step"])["kernel"]
    # the scene decides which class dominates: the group gather on the two Cornell scenes; through 10 000 spheres the two
    # emissions (two launches per step) or the shadow pass
    if scene == "spheres10k":
        assert roof["kernel"] in ("k_emit_stream", "k_shadow_rays"), roof["kernel"]
    else:
        assert roof["kernel"] == "k_gather_tile", roof["kernel"]
        assert isinstance(roof["frac"], float) and 0 < roof["frac"] < 1, roof       # a number with or without counters
    inter = j["intersection"]
    assert inter["k_primary"]["plane_tests"] == 5 * 1440 * 1440 and inter["k_shadow_rays"]["sphere_tests"] > 0
    if scene == "default":
        assert inter["k_primary"]["sphere_tests"] == 2 * 1440 * 1440          # the linear scan: every ray tests every primitive
        assert roof["bound"] == "valu" and roof["traffic"] and roof["traffic"] > 0, roof   # counters collected in this very run
        assert 0 < roof["hbm"]["frac"] < 1
        assert j["per_query_gather"]["valu_instr_per_gather"] > 0
        for k in ("k_primary", "k_shadow_rays"):                              # north_star: VALU / occupancy for intersection
            assert 0 < inter[k]["waves_per_simd_achieved"] <= 8.5 and 0 < inter[k]["valu_issue_frac"] < 1.2, inter[k]
    else:
        assert roof["traffic"] is None
        if scene == "caustics":
            assert roof["bound"] == "hbm" and "fallback" in roof                # no counters: SURVEY 8d's figure stands in, named as such
