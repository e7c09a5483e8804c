"""Per-queue kernel timeline of Groth16 proofs from a rocprofv3 kernel trace (consecutive kernels of one name merged):
python tools/dbg/trace_g16.py dir [start_us] [span_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
start = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
span = float(sys.argv[3]) if len(sys.argv) > 3 else 8000.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)
    m = re.match(r'([a-zA-Z0-9_]+)(<[^(]*)?', n)
    base = m.group(1)[:22]
    if 'Fp2' in n: base += '<G2>'
    return base
qc = [r for r in rows if 'k_qap_combine' in r['Kernel_Name']]
base = int(qc[len(qc) * 2 // 3]['Start_Timestamp']) - 3000000      # ~3 ms before a late proof's h step
merged = []
for r in rows:
    s, e = (int(r['Start_Timestamp']) - base) / 1e3, (int(r['End_Timestamp']) - base) / 1e3
    if s < start or s > start + span: continue
    key = (r['Queue_Id'], nm(r))
    if merged and merged[-1][0] == key and s - merged[-1][2] < 60: merged[-1][2] = e; merged[-1][3] += 1
    else: merged.append([key, s, e, 1])
for (q, n), s, e, c in merged:
    print(f"q{q} {n:28s} {s:9.1f} -> {e:9.1f}  ({e - s:7.1f} us) x{c}")
