"""Per-kernel means of any rocprofv3 --pmc pass:  python tools/dbg/pmc_table.py <dir> [kernel substring ...]"""
import csv, glob, os, re, collections, sys
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"kg::", "", n)
    if sys.argv[2:] and not any(a in n for a in sys.argv[2:]): continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n][r["Counter_Name"]] += 1
for n, c in acc.items():
    print(f"{n[:60]:60s} x{max(cnt[n].values()):4d} " + "  ".join(f"{x}={v / cnt[n][x]:.4g}" for x, v in c.items()))
