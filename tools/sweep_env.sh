#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]   -> ms per step of bench.py for each value of an environment switch
var=$1; vals=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/sweep
for v in $vals; do
  export $var=$v
  timeout -k 10 300 python3 bench.py --no-pmc --cpu-row-stride 16 --check "$@" 2> $GRAFT_REPO_ROOT/gpurun_out/sweep/err_$v.txt | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['phases_ms_rank0']
print('$var=$v', 'ms/step', round(j['ms_per_step'],3), 'gather', round(p['ms_gather'],3), 'groups g/c', round(p['ms_group_global'],3), round(p['ms_group_caustic'],3), 'identical', j['cpu_baseline']['sample_rows_bit_identical_to_gpu'])" || { tail -3 $GRAFT_REPO_ROOT/gpurun_out/sweep/err_$v.txt; exit 1; }
done
