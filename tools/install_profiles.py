"""Copies what tools/collect_profiles.sh left under gpurun_out/collect into profiles/<round>_*:
    python tools/install_profiles.py r02"""
import glob, json, os, shutil, subprocess, sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "collect")
dst = os.path.join(root, "profiles")


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


for name, out in (("bench.json", f"{rnd}_bench.json"), ("bench_under_rocprof.json", f"{rnd}_bench_under_rocprof.json"),
                  ("msm_under_rocprof.json", f"{rnd}_msm_under_rocprof.json")):
    json.dump(last_json_line(os.path.join(src, name)), open(os.path.join(dst, out), "w"), indent=1)
for d, out in (("kt_bench", f"{rnd}_bench_kernel_stats.csv"), ("kt_msm", f"{rnd}_msm_kernel_stats.csv")):
    f = glob.glob(os.path.join(src, d, "*", "*kernel_stats.csv"))[0]
    shutil.copy(f, os.path.join(dst, out))
subprocess.check_call([sys.executable, os.path.join(root, "tools", "dbg", "pmc_summary.py"), os.path.join(src, "pmc_"),
                       os.path.join(dst, f"{rnd}_pmc_hbm.json")])
print("installed into", dst)
