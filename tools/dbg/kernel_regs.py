"""VGPR / scratch / LDS / occupancy of the kernels of one source, as the compiler reports them:
    python tools/dbg/kernel_regs.py msm.hip [name substring ...] [-- extra hipcc flags]"""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "..", "kogarashi_amd", "csrc")
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src, subs = args[0], args[1:]
cmd = ["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off", "--cuda-device-only", "-w",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, src), "-o", "/dev/null", "-I", os.path.join(here, "..", "..", "include")] + extra
t = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in t.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    else:
        cur[k.split(" ")[0]] = v
    if k.startswith("LDS"):
        dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem); dem = re.sub(r"\(.*", "", dem); dem = re.sub(r"kg::", "", dem)
        if subs and not any(a in dem for a in subs): continue
        print(f"{dem[:72]:72s} vgpr {cur.get('VGPRs','?'):>4s} sgpr {cur.get('SGPRs','?'):>4s} scratch {cur.get('ScratchSize','?'):>4s} occ {cur.get('Occupancy','?')}")
