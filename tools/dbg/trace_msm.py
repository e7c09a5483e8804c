"""Per-queue kernel timeline of the LAST blocking MSM of a rocprofv3 kernel trace (from its k_prep_scalars* launch on):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/dbg/groups_timeline.py 20
   python tools/dbg/trace_msm.py gpurun_out/x"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)
    m = re.match(r'([A-Za-z_0-9]+)(<.*?>)?\(', n)
    k = m.group(1) if m else n[:30]
    if 'Fp2' in n.split('(')[0]: k += '<G2>'
    return k
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r), r['Queue_Id'], r.get('Grid_Size', r.get('Grid_Size_X', '?'))) for r in csv.DictReader(open(f)))
starts = [i for i, e in enumerate(ev) if e[2].startswith('k_prep_scalars')]
i0 = starts[-1]
while i0 > 0 and ev[i0 - 1][2] == 'k_zero' and ev[i0][0] - ev[i0 - 1][1] < 50000: i0 -= 1
t0 = ev[i0][0]
prev_end = {}
for e in ev[i0:]:
    gap = (e[0] - prev_end[e[3]]) / 1e3 if e[3] in prev_end else 0.0
    prev_end[e[3]] = e[1]
    print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:8.1f}  gap {gap:7.1f}  q{e[3]:>3s} {e[2]:24s} grid {e[4]}")
