"""Kernel timeline of the last whole step of a traced bench run: start, end and duration of every launch relative to the
step's first one, with its hardware queue, from rocprofv3's kernel trace.  (Under the tracer a launch costs the host
several times what it costs otherwise, so the streams start later than they do in an untraced run.)
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -o tr -- python3 bench.py --steps 3 --no-cpu-baseline
       python3 tools/timeline.py <dir>/tr_kernel_trace.csv [minimum microseconds to show]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
floor_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]      # the last kernel of a step
if len(ends) < 2:
    sys.exit("need a trace of at least two steps")
b, e = ends[-2] + 1, ends[-1]
t0 = int(rows[b]["Start_Timestamp"])
for r in rows[b:e + 1]:
    s, t = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if t - s < floor_us:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("pm::", "")
    print(f"{s:9.1f} {t:9.1f} {t - s:8.1f} us  q{r['Queue_Id']} {name[:60]}")
