#!/bin/bash
# kernel trace + stats of the profiling driver
out=${1:-gpurun_out/trace}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o tr -- python3 $R/tools/prof_frame.py 1000000 ${2:-5} > $R/$out.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/$out/tr_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
