"""Diagnostics: timeline of the kernels of the last bench step from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last k_emit pair back
emits = [i for i, r in enumerate(rows) if "k_emit<" in r["Kernel_Name"]]
start = emits[-2]
t0 = int(rows[start]["Start_Timestamp"])
agg = {}
for r in rows[start:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    q = r.get("Queue_Id", "?")
    if "rocprim" in name or "boxes_up" in name:
        k = (q, "sort/boxes_up")
        a = agg.setdefault(k, [s, e, 0]); a[1] = e; a[2] += 1
        continue
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f}  q{q} {name}")
for (q, n), (s, e, c) in agg.items():
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f}  q{q} {n} x{c}")
