"""average the counters of the last N dispatches of a kernel (name substring) in a rocprofv3 --pmc CSV"""
import csv, glob, os, sys
d, sub = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-last:]
names = sorted({n for i in ids for n in by[i]})
print("  ".join("%s=%.4g" % (n, sum(by[i].get(n, 0) for i in ids) / len(ids)) for n in names))
