"""summarise a rocprofv3 --pmc pass of bench.py per search dispatch of the last isolated launches"""
import csv, glob, os, sys
d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "ph_search_kernel" in r["Kernel_Name"]]
names = sorted({r["Counter_Name"] for r in rows})
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-10 * per:]
for ph in range(per):
    sel = [by[i] for k, i in enumerate(ids) if k % per == ph]
    print("dispatch %d of %d:" % (ph, per), "  ".join("%s=%.4g" % (n, sum(s.get(n, 0) for s in sel) / len(sel)) for n in names))
