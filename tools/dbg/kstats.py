"""Prints a rocprofv3 kernel_stats.csv compactly:  python tools/dbg/kstats.py gpurun_out/<dir>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    name = re.sub(r'(kg::)?(msm::)?\(anonymous namespace\)::|kg::msm::|kg::', '', r['Name'])
    name = re.sub(r'\(.*', '', name)[:50]
    print(f"{name:50s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
