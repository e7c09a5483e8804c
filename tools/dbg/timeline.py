"""One Groth16 proof (between the last two k_qap_combine launches) as a per-queue timeline from a rocprofv3 kernel trace.
python tools/dbg/timeline.py gpurun_out/<dir> [min_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)
    m = re.match(r'([A-Za-z_0-9]+)(<.*?>)?\(', n)
    k = m.group(1) if m else n[:30]
    if 'Fp2' in n.split('(')[0]: k += '<G2>'
    return k
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r), r['Queue_Id']) for r in csv.DictReader(open(f)))
q = [i for i, e in enumerate(ev) if e[2] == 'k_qap_combine']
i0, i1 = q[-2], q[-1]
t0 = ev[i0][0]
print("period us:", (ev[i1][0] - t0) / 1e3)
for e in ev[i0:i1 + 1]:
    if (e[1] - e[0]) / 1e3 >= min_us:
        print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:8.1f}  q{e[3]} {e[2]}")
