#!/bin/bash
# Round 3's evidence, on the GPU box: tools/round3_evidence.sh <part>   (parts: rates, default, scenes, config3, ranks, waves)
# Everything lands under gpurun_out/ev3/ ; what is to be kept is copied into profiles/r03/ afterwards.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ev3
mkdir -p $out
cd $R
case "$1" in
rates)
  # what a vector instruction costs with every SIMD busy, and the clock it ran at (rocm-smi before / during is not available to
  # an ordinary user on every box: GRBM counters of a frame give the busy clock)
  timeout -k 10 120 tools/microbench/valu_rates > $out/valu_rates.txt 2>&1 || { tail -3 $out/valu_rates.txt; exit 1; }
  (rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -4) >> $out/valu_rates.txt
  cat $out/valu_rates.txt | head -12 ;;
default)
  bash tools/profile_bench.sh ev3_default && cp -r gpurun_out/prof_ev3_default $out/default
  timeout -k 10 400 python3 bench.py --check > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; } ;;
scenes)
  for s in caustics spheres10k; do
    bash tools/profile_bench.sh ev3_$s --scene $s && cp -r gpurun_out/prof_ev3_$s $out/$s
    timeout -k 10 600 python3 bench.py --scene $s --check --cpu-row-stride 48 > $out/bench_$s.json 2> $out/bench_$s.err || { tail -5 $out/bench_$s.err; exit 1; }
  done ;;
config3)
  # BASELINE config 3's sizes on one GPU: pane 7680 x 4320 (3840 x 2160 at 4 spp), 10M + 10M photons
  bash tools/profile_bench.sh ev3_config3 --width 7680 --height 4320 --photons 10000000 && cp -r gpurun_out/prof_ev3_config3 $out/config3
  timeout -k 10 900 python3 bench.py --width 7680 --height 4320 --photons 10000000 --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_config3.json 2> $out/bench_config3.err || { tail -5 $out/bench_config3.err; exit 1; } ;;
ranks)
  timeout -k 10 300 python3 tools/rank_share.py > $out/rank_share.txt 2>&1; tail -8 $out/rank_share.txt ;;
waves)
  PM_LIB=$R/ab/lib_wt.so timeout -k 10 120 python3 tools/tile_waves.py > $out/tile_waves.txt 2>&1; cat $out/tile_waves.txt | grep -v amdgpu.ids ;;
esac
