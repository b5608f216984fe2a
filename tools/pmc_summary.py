"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one row per dispatch x counter)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].replace("slamx::(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(", ", ",").replace(",", ";")  # template arguments: no commas in a CSV field
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
names = sorted({r["Counter_Name"] for r in rows})
print("kernel,launches," + ",".join(names))
for k in acc:
    n = len(cnt[k])
    print(k + "," + str(n) + "," + ",".join("%.4g" % (acc[k][c] / n) for c in names))
