#!/bin/bash
# rocprofv3 evidence for bench.py's numbers: kernel trace + stats, then PMC passes (one rocprofv3 run per counter set, as
# MI355X_MICROARCH.md prescribes), all on the same command.  Writes gpurun_out/prof_<tag>/ ; copy what is to be kept
# into profiles/.
# usage: profile_bench.sh <tag> [extra bench.py arguments, e.g. --scene spheres10k]
tag=${1:-r01}
shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o tr -- $CMD > $out/trace.log 2>&1 || { echo "trace run failed"; tail -5 $out/trace.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --output-format csv --pmc $set -d $out/p$i -o pmc -- $CMD > $out/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 $out/p5 $out/p6 > $out/pmc_per_dispatch_avg.json
cp $out/trace/tr_kernel_stats.csv $out/kernel_stats_bench.csv
grep '^{' $out/trace.log | tail -1 > $out/bench_under_trace.json
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats_bench.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
