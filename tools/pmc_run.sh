#!/bin/bash
# PMC passes over the profiling driver (one rocprofv3 run per counter set; see MI355X_MICROARCH.md)
set -e
out=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc $set -d $R/$out/p$i -o pmc -- python3 $R/tools/prof_frame.py 1000000 2 > $R/$out.p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/$out > $R/$out.json
