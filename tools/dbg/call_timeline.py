import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n); n=re.sub(r'kg::msm::|kg::','',n)
    return re.match(r'([a-zA-Z0-9_]+)', n).group(1)
key = sys.argv[2] if len(sys.argv) > 2 else 'k_prep_scalars'
idx=[i for i,r in enumerate(rows) if key in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
base=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s,e=(int(r['Start_Timestamp'])-base)/1e3,(int(r['End_Timestamp'])-base)/1e3
    print(f"q{r['Queue_Id']:>3} {nm(r):26s} {s:8.1f} -> {e:8.1f} ({e-s:6.1f})")
print("next call starts at", (int(rows[b]['Start_Timestamp'])-base)/1e3)
