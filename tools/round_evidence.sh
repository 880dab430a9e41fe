#!/bin/bash
# the round's bench lines: default bench command, then the two other scenes with the CPU sample rows checked
# usage (on the GPU box): tools/round_evidence.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ev_$tag
mkdir -p $out
cd $R
timeout -k 10 400 python3 bench.py --check > $out/bench_default.json 2> $out/bench_default.err || { echo "default failed"; tail -5 $out/bench_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --scene caustics --check > $out/bench_caustics.json 2> $out/bench_caustics.err || { echo "caustics failed"; tail -5 $out/bench_caustics.err; exit 1; }
timeout -k 10 600 python3 bench.py --scene spheres10k --check --cpu-row-stride 48 > $out/bench_spheres10k.json 2> $out/bench_spheres10k.err || { echo "spheres10k failed"; tail -5 $out/bench_spheres10k.err; exit 1; }
python3 - <<PY
import json
for s in ("default", "caustics", "spheres10k"):
    j = json.loads(open("$out/bench_%s.json" % s).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(s, "ms/step", round(j["ms_per_step"], 3), "Mrays/s", round(j["value"], 1), "roofline", r["kernel"], r["bound"], round(r["frac"], 4), "cpu", j.get("cpu_baseline", {}).get("value"), "check", j.get("check"))
PY
