"""Kernel timeline of the last step of a traced bench run: start and end of every launch relative to the step's first
kernel, from rocprofv3's kernel trace (tr_kernel_trace.csv).  usage: timeline.py <kernel_trace.csv> [launches per step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the emission of the caustic map: find the last k_emit launch pair
starts = [i for i, r in enumerate(rows) if "k_emit<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("void pm::k_emit")]
# group launches into steps: a new step when an emission kernel follows a k_resolve
step_begin = [i for i in starts if i == 0 or any("k_resolve" in rows[j]["Kernel_Name"] for j in range(max(0, i - 40), i)) and
              not any("k_emit" in rows[j]["Kernel_Name"] for j in range(max(0, i - 3), i))]
b = step_begin[-1]
t0 = int(rows[b]["Start_Timestamp"])
for r in rows[b:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("void ", "").replace("pm::", "")
    if e - s < 15 and "gather" not in name:
        continue
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  q{r.get('Queue_Id', '?'):>3s} {name[:60]}")
