"""The last `span` us of a rocprofv3 kernel trace, per queue: python tools/dbg/tail_timeline.py dir [span_us] [end_offset_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
off = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
end = max(int(r['End_Timestamp']) for r in rows) - off * 1e3
base = end - span * 1e3
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n); n = re.sub(r'kg::msm::|kg::', '', n)
    m = re.match(r'([a-zA-Z0-9_]+)', n)
    return m.group(1)[:24] + ('<G2>' if 'Fp2' in n else '')
for r in rows:
    s, e = (int(r['Start_Timestamp']) - base) / 1e3, (int(r['End_Timestamp']) - base) / 1e3
    if s < 0 or s > span: continue
    print(f"q{r['Queue_Id']:>3} {nm(r):30s} {s:8.1f} -> {e:8.1f} ({e - s:6.1f})")
