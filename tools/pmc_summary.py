"""Average per-dispatch PMC values per kernel from rocprofv3 counter_collection CSVs found under the given directories."""
import sys, csv, glob, json, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        with open(path) as f:
            for row in csv.DictReader(f):
                per[(row["Kernel_Name"].split("(")[0], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (k, _, c), v in per.items():
            acc[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
print(json.dumps(out, indent=1))
