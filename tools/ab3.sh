#!/bin/bash
# several library builds in turn, twice: tools/ab3.sh lib1 lib2 ... ("-" = the in-tree build); optional bench args via BENCH_ARGS
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset PM_LIB; else export PM_LIB=$PWD/$lib; fi
    timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc $BENCH_ARGS --check-hash 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); p=d['phases_ms_rank0']
print('$lib'.ljust(22), 'step', round(d['ms_per_step'],3), 'total', round(p['ms_total'],3), 'gather', round(p['ms_gather'],3), 'g', round(p['ms_gather_global'],3), 'c', round(p['ms_gather_caustic'],3), 'groups', round(p['ms_group_global'],3), round(p['ms_group_caustic'],3), d.get('pane_sha1','')[:12])" || exit 1
  done
done
