"""k_acc_tasks wave-cycle breakdown from a rocprofv3 --pmc SQ_* pass:  python tools/dbg/pmc_wait.py dir"""
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"[<(].*", "", n)
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1
for n in ("k_acc_tasks", "k_group_scatter", "k_halve"):
    c = acc[n]; k = cnt[n] or 1
    print(n, "x", k, {x: round(v / k / 1e6, 1) for x, v in c.items()})
