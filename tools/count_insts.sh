#!/bin/bash
# vector / scalar instructions per launch of every kernel of a short bench run (one PMC pass); prints the gather kernels
# usage: count_insts.sh [bench.py arguments]
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/insts
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $out/p -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc $* > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 $R/tools/pmc_summary.py $out/p > $out/insts.json
python3 - <<PY
import json
d = json.load(open("$out/insts.json"))
for k, v in d.items():
    if "gather" in k or "emit<" in k or "primary" in k or "shadow" in k:
        print(f"{k[:48]:48s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:8.2f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:8.2f} M")
PY
