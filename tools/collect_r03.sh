#!/bin/bash
# copy what is to be judged from gpurun_out/ev3 (tools/round3_evidence.sh) into profiles/r03, and print a summary
cd "$(dirname "$0")/.."
E=gpurun_out/ev3
for s in default caustics spheres10k config3; do
  cp $E/$s/kernel_stats_bench.csv profiles/r03/${s}_kernel_stats_bench.csv; cp $E/$s/pmc_per_dispatch_avg.json profiles/r03/${s}_pmc_per_dispatch_avg.json
  cp $E/$s/bench_under_trace.json profiles/r03/${s}_bench_under_trace.json; cp $E/bench_$s.json profiles/r03/bench_$s.json
done
cp $E/default/kernel_stats_bench.csv profiles/r03/kernel_stats_bench.csv; cp $E/default/pmc_per_dispatch_avg.json profiles/r03/pmc_per_dispatch_avg.json
grep -v amdgpu.ids $E/rank_share.txt > profiles/r03/rank_share.txt; cp $E/rank_share_config3.txt profiles/r03/rank_share_config3.txt
[ -f gpurun_out/tw8.txt ] && cp gpurun_out/tw8.txt profiles/r03/tile_waves_rank0_of_8.txt; [ -f gpurun_out/tw1.txt ] && cp gpurun_out/tw1.txt profiles/r03/tile_waves.txt
python3 tools/timeline.py $E/default/trace/tr_kernel_trace.csv 0 > profiles/r03/default_timeline_traced_step.txt
python3 tools/timeline.py $E/config3/trace/tr_kernel_trace.csv 100 > profiles/r03/config3_timeline_traced_step.txt
python3 - <<'P'
import json, csv
for s in ['default','caustics','spheres10k','config3']:
    d=json.load(open('profiles/r03/bench_%s.json'%s))
    r=d['roofline']
    print(s, round(d['ms_per_step'],3), round(d['value']), round(d['gathers_per_s']/1e6), 'host pane', round(d['ms_per_step_host_pane'],2), '|', r['kernel'], round(r['ms_per_launch'],3), 'frac', round(r['frac'],3), 'hbm', r.get('hbm',{}).get('frac'), 'cpu', d.get('cpu_baseline',{}).get('value'))
    for x in d['roofline_all']:
        print('    ', x['kernel'], round(x['ms_per_launch'],3), 'valu', round(x['valu_issue']['frac'],3), 'instr', round(x['valu_issue']['valu_instr_per_launch']/1e6,1), 'occ', round(x['occupancy']['waves_per_simd_achieved'],2) if x.get('occupancy') else None, 'traffic', x.get('traffic'))
    print('    inter', {k:(round(v['ms'],3), round(v['frac_of_fp64_peak'],4), round(v.get('waves_per_simd_achieved',0),2)) for k,v in d['intersection'].items()})
    print('    pq', round(d['per_query_gather']['valu_instr_per_gather']), d['per_query_gather']['gathers_per_step'], {k:round(v,3) for k,v in d['phases_ms_rank0'].items() if k.startswith('map_') or k in ('ms_total','ms_gather')})
    rows=list(csv.DictReader(open('profiles/r03/%s_kernel_stats_bench.csv'%s)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print('    trace:', ', '.join('%s %.0f us x%s (%.1f%%)'%(r['Name'].split('(')[0].replace('void pm::','').replace('pm::','')[:18], float(r['AverageNs'])/1e3, r['Calls'], 100*float(r['TotalDurationNs'])/tot) for r in rows[:7]))
P
grep world profiles/r03/rank_share.txt; grep world profiles/r03/rank_share_config3.txt
