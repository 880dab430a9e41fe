#!/bin/bash
# A/B/A/B of library builds on one box: tools/ab_run.sh <outtag> <libA|-> <libB|-> [bench.py arguments]   ("-" = the in-tree build)
# prints ms_per_step of each run; the bench lines go to gpurun_out/<outtag>/
tag=$1; A=$2; B=$3; shift; shift; shift
out=gpurun_out/$tag; mkdir -p $out
for round in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    if [ "$lib" = "-" ]; then unset PM_LIB; else export PM_LIB=$PWD/$lib; fi
    timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc "$@" > $out/$v$round.json 2> $out/$v$round.err || { echo "run $v$round failed"; tail -5 $out/$v$round.err; exit 1; }
    python3 - $out/$v$round.json $v$round "$lib" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["phases_ms_rank0"]
print(f"{sys.argv[2]} {sys.argv[3]:24s} step {d['ms_per_step']:.3f} ms  total {p['ms_total']:.3f} gather {p['ms_gather']:.3f} (g {p['ms_gather_global']:.3f} c {p['ms_gather_caustic']:.3f}; groups {p['ms_group_global']:.3f} {p['ms_group_caustic']:.3f})")
PY
  done
done
