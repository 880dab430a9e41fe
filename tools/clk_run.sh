cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/clk -o pmc -- python3 $R/tools/prof_frame.py 1000000 3 > $R/gpurun_out/clk.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/clk
