"""Per-kernel instruction counts from a rocprofv3 --pmc SQ_INSTS_* pass:  python tools/dbg/pmc_insts.py dir"""
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"[<(].*", "", n)
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[n] += 1
for n, c in acc.items():
    k = cnt[n] or 1
    g = lambda x: c[x] / k / 1e6
    print(f"{n:26s} x{k:4d} VALU {g('SQ_INSTS_VALU'):10.2f}M SALU {g('SQ_INSTS_SALU'):8.2f}M LDS {g('SQ_INSTS_LDS'):8.2f}M "
          f"VMEMrd {g('SQ_INSTS_VMEM_RD'):8.2f}M wr {g('SQ_INSTS_VMEM_WR'):7.2f}M waves {c['SQ_WAVES'] / k:9.0f}")
