"""Per-kernel table of a rocprofv3 --kernel-trace run, from the per-launch trace (not the stats file, which averages every launch of a name):

    python tools/kernel_table.py <rocprof output dir> [--per N] [--split-grid NAME] [--last-of NAME K] [--title TEXT]

  --per N           also print the microseconds per unit of work: total / N (N = proofs, MSMs ... the run produced)
  --split-grid NAME launches of kernels whose short name contains NAME are split by grid size (the prover launches k_acc_tasks<Fq> twice per
                    proof: the fused a / b_g1 / l accumulation and h's -- different grids)
  --last-of NAME K  a second average over only the LAST K launches of the kernels whose short name contains NAME (the timed region of
                    `bench.py --headline-only`: its pre-warm and warm-up launches come first)

Families are the kernels' names without namespaces and argument lists; G2 instances (Fp2 / Fp2S template arguments) are marked.  The trace's
kernel time is End - Start per launch; launches on different queues overlap, so the per-unit sums exceed the wall time per unit."""
import argparse
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"kg::msm::|kg::", "", n)
    m = re.match(r"([A-Za-z0-9_]+)(<.*?>)?\(", n + "(")
    base = m.group(1) if m else n[:40]
    targs = n[len(base):].split("(")[0]
    tag = ""
    if "Fp2" in targs:
        tag = "<G2>"
    elif "FrParams" in targs and "Fp<" in targs:
        tag = "<Fr>"
    elif "FqParams" in targs and "Fp<" in targs:
        tag = "<Fq>"
    m2 = re.match(r"<(\d+(?:, ?\d+)*)", targs)
    if m2 and not tag:
        tag = "<" + m2.group(1).replace(" ", "") + ">"
    return base + tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--per", type=float, default=0.0)
    ap.add_argument("--split-grid", default="")
    ap.add_argument("--last-of", nargs=2, default=None)
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "*", "*kernel_trace.csv")) + glob.glob(os.path.join(a.dir, "*kernel_trace.csv"))
    if not files:
        sys.exit(f"no kernel_trace.csv under {a.dir}")
    rows = sorted(csv.DictReader(open(max(files, key=os.path.getmtime))), key=lambda r: int(r["Start_Timestamp"]))
    fam = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        if a.split_grid and a.split_grid in k:
            k += f" grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}"
        fam.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if a.title:
        print(a.title)
    tot_all = sum(sum(v) for v in fam.values())
    hdr = f"{'kernel':46s} {'launches':>8s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'total ms':>9s} {'share':>6s}"
    if a.per:
        hdr += f" {'us per unit':>12s}"
    print(hdr)
    for k, v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        line = f"{k[:46]:46s} {len(v):8d} {sum(v) / len(v):10.1f} {min(v):9.1f} {max(v):9.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot_all:5.1f}%"
        if a.per:
            line += f" {sum(v) / a.per:12.1f}"
        print(line)
    print(f"{'all kernels':46s} {sum(len(v) for v in fam.values()):8d} {'':10s} {'':9s} {'':9s} {tot_all / 1e3:9.2f}" + (f" {'':6s} {tot_all / a.per:12.1f}" if a.per else ""))
    if a.last_of:
        name, kk = a.last_of[0], int(a.last_of[1])
        for k, v in fam.items():
            if name in k and len(v) >= kk:
                w = v[-kk:]
                print(f"last {kk} launches of {k}: avg {sum(w) / kk:.1f} us (min {min(w):.1f}, max {max(w):.1f}); all {len(v)}: avg {sum(v) / len(v):.1f} us")


if __name__ == "__main__":
    main()
