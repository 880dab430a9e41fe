#!/bin/bash
# usage: tools/sweep_fast.sh VAR "v1 v2 ..." [bench args]   -> ms per step for each value of an environment switch, twice, with the pane's hash
var=$1; vals=$2; shift 2
for round in 1 2; do
for v in $vals; do
  export $var=$v
  timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --check-hash "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms_rank0']
print('$var=$v', 'ms/step', round(j['ms_per_step'],3), 'gather', round(p['ms_gather'],3), 'g/c', round(p['ms_gather_global'],3), round(p['ms_gather_caustic'],3), 'groups g/c', round(p['ms_group_global'],3), round(p['ms_group_caustic'],3), j['pane_sha1'][:10])" || exit 1
done
done
