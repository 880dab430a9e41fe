#!/bin/bash
# usage: tools/build_alone.sh <tag> [library] [photons]   -> per-kernel averages of 16 lone map builds
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/ba_$1; mkdir -p $out
[ -n "$2" ] && export PM_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o tr -- python3 $R/tools/build_alone.py ${3:-1000000} > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/trace/tr_kernel_stats.csv")))
tot = 0
for r in rows:
    if "emit" in r["Name"] or "scan_blocks" in r["Name"]: continue
    per_build = float(r["TotalDurationNs"]) / 16 / 1e3
    tot += per_build
    print(f"{r['Name'][:50]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  per build {per_build:8.1f} us")
print("sum per build (without emission)", round(tot, 1), "us")
PY
