"""Prints the kernels of a rocprofv3 kernel trace that start inside [t0, t1] us after the first k_acc_tasks dispatch of the
last third of the run:  python tools/dbg/trace_window.py dir [span_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 3500.0
rows = list(csv.DictReader(open(f)))
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)
    return re.sub(r'[<(].*', '', n)[:28]
accs = [r for r in rows if 'k_acc_tasks' in r['Kernel_Name']]
base = int(accs[len(accs) * 2 // 3]['Start_Timestamp'])
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    s, e = (int(r['Start_Timestamp']) - base) / 1e3, (int(r['End_Timestamp']) - base) / 1e3
    if -400 <= s <= span and (e - s > 4 or 'halve' not in r['Kernel_Name']):
        print(f"q{r['Queue_Id']} {nm(r):28s} {s:9.1f} -> {e:9.1f}  ({e - s:7.1f} us)")
