#!/bin/bash
# kernel timeline of the last step of a short traced bench run -> gpurun_out/<tag>/timeline.txt
# usage: timeline_run.sh <tag> [minimum microseconds] [bench.py arguments]
tag=${1:-tl}; floor=${2:-0}; shift; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o tr -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc $* > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python3 $R/tools/timeline.py $out/trace/tr_kernel_trace.csv $floor > $out/timeline.txt
rm -rf $out/trace
cat $out/timeline.txt
