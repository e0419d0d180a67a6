"""Generic per-kernel summary of a rocprofv3 --pmc CSV: python tools/pmc_by_kernel.py <dir> [substring filter]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0][:100]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -len(cnt[kv[0]])):
    n = len(cnt[k])
    print(k, "launches", n, {c: round(x / n) for c, x in sorted(v.items())})
