"""Copies what tools/collect_profiles.sh left under gpurun_out/collect into profiles/<round>_*:
    python tools/install_profiles.py r02"""
import glob, json, os, shutil, subprocess, sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "collect")
dst = os.path.join(root, "profiles")


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: an earlier collection's files stay beside the latest"""
    files = glob.glob(pattern)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return max(files, key=os.path.getmtime)


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def sq_counters(pass_dir, out_path, only=None):
    """per-kernel means of a rocprofv3 --pmc SQ_* pass"""
    import csv, collections, re
    f = newest(os.path.join(pass_dir, "*", "*counter_collection.csv"))
    acc, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if only and only not in name:
            continue
        if "k_ntt_tile" in name:
            key = re.search(r"k_ntt_tile<[^>]*>", name).group(0).replace(" ", "") + "/grid" + r["Grid_Size"]
        else:
            key = re.sub(r"(kg::)?(msm::)?\(anonymous namespace\)::|kg::msm::|kg::", "", name); key = re.sub(r"^void ", "", key); key = re.sub(r"\(.*", "", key)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key] += r["Counter_Name"] == "SQ_WAVE_CYCLES"
    out = {k: {"dispatches": cnt[k], **{c: v / max(cnt[k], 1) for c, v in d.items()}} for k, d in acc.items()}
    out["note"] = ("per dispatch; SQ_* in quad-cycles summed over all waves (SQ_WAIT_ANY: parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY: waiting "
                   "to issue, SQ_ACTIVE_INST_ANY: issuing); GRBM_GUI_ACTIVE summed over the 8 XCDs; counter passes serialise the dispatches")
    json.dump(out, open(out_path, "w"), indent=1)


for name, out in (("bench.json", f"{rnd}_bench.json"), ("bench_under_rocprof.json", f"{rnd}_bench_under_rocprof.json"),
                  ("msm_under_rocprof.json", f"{rnd}_msm_under_rocprof.json")):
    json.dump(last_json_line(os.path.join(src, name)), open(os.path.join(dst, out), "w"), indent=1)
for d, out in (("kt_bench", f"{rnd}_bench_kernel_stats.csv"), ("kt_msm", f"{rnd}_msm_kernel_stats.csv")):
    f = newest(os.path.join(src, d, "*", "*kernel_stats.csv"))
    shutil.copy(f, os.path.join(dst, out))
subprocess.check_call([sys.executable, os.path.join(root, "tools", "dbg", "pmc_summary.py"), os.path.join(src, "pmc_"),
                       os.path.join(dst, f"{rnd}_pmc_hbm.json")])
# the transform alone: kernel averages, HBM counters, SQ issue / wait counters
if os.path.isdir(os.path.join(src, "kt_ntt")):
    shutil.copy(newest(os.path.join(src, "kt_ntt", "*", "*kernel_stats.csv")), os.path.join(dst, f"{rnd}_ntt_kernel_stats.csv"))
    json.dump(last_json_line(os.path.join(src, "ntt_under_rocprof.json")), open(os.path.join(dst, f"{rnd}_ntt_under_rocprof.json"), "w"), indent=1)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "dbg", "pmc_summary.py"), os.path.join(src, "pmcntt_"),
                           os.path.join(dst, f"{rnd}_ntt_pmc_hbm.json")])
    sq_counters(os.path.join(src, "pmcntt_SQ"), os.path.join(dst, f"{rnd}_ntt_sq_counters.json"), only="k_ntt_tile")
if os.path.isdir(os.path.join(src, "pmc_SQ")):
    sq_counters(os.path.join(src, "pmc_SQ"), os.path.join(dst, f"{rnd}_msm_sq_counters.json"))
# round 6: the prover alone, the G2 MSM alone, the three other transforms, and the per-launch tables (tools/kernel_table.py)
for d, out in (("kt_g16", "groth16"), ("kt_g2", "msm_g2"), ("kt_ntt_idft", "ntt_idft"), ("kt_ntt_coset_dft", "ntt_coset_dft"), ("kt_ntt_coset_idft", "ntt_coset_idft")):
    if os.path.isdir(os.path.join(src, d)):
        shutil.copy(newest(os.path.join(src, d, "*", "*kernel_stats.csv")), os.path.join(dst, f"{rnd}_{out}_kernel_stats.csv"))
for name, out in (("g16_under_rocprof.json", "groth16_under_rocprof.json"), ("g2_under_rocprof.json", "msm_g2_under_rocprof.json"),
                  ("ntt_idft_under_rocprof.json", "ntt_idft_under_rocprof.json"), ("ntt_coset_dft_under_rocprof.json", "ntt_coset_dft_under_rocprof.json"),
                  ("ntt_coset_idft_under_rocprof.json", "ntt_coset_idft_under_rocprof.json")):
    if os.path.exists(os.path.join(src, name)):
        json.dump(last_json_line(os.path.join(src, name)), open(os.path.join(dst, f"{rnd}_{out}"), "w"), indent=1)
for name, out in (("msm_kernel_table.txt", "msm_kernel_table.txt"), ("g16_phase_table.txt", "groth16_phase_table.txt"), ("g2_kernel_table.txt", "msm_g2_kernel_table.txt")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{rnd}_{out}"))
for name in ("mul_rate.txt", "ntt_pass_rate.txt", "mfma_const_mul.txt", "group_scatter.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{rnd}_{name}"))
print("installed into", dst)
