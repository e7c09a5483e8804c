"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) into one
JSON: per kernel the mean KB per dispatch, plus VGPR / LDS use.  python tools/dbg/pmc_summary.py gpurun_out/pmc_ out.json"""
import csv, glob, json, re, sys, collections
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    import os
    f = max(glob.glob(f"{sys.argv[1]}{counter}/*/*counter_collection.csv"), key=os.path.getmtime)      # gpurun_out keeps earlier collections' files
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"(kg::)?(msm::)?\(anonymous namespace\)::|kg::msm::|kg::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
        key = m.group(1) + (m.group(2) if m and m.group(2) and ("Fp2" in m.group(2)) else "") if m else name[:40]
        if m and m.group(2) and "Fp2" in m.group(2):
            key = m.group(1) + "<Fq2>"
        if m and m.group(1) == "k_ntt_tile" and m.group(2):
            key = "k_ntt_tile" + m.group(2).replace(" ", "") + "/grid" + r.get("Grid_Size", "?")     # one key per tile shape and transform size
        a = acc.setdefault(key, {"dispatches": 0, "sum": 0.0, "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "wg": int(r["Workgroup_Size"])})
        a["dispatches"] += 1
        a["sum"] += float(r["Counter_Value"])
    out[counter] = {k: {"dispatches": v["dispatches"], "mean_KB": round(v["sum"] / v["dispatches"], 1), "vgpr": v["vgpr"], "lds_bytes": v["lds"], "workgroup": v["wg"]} for k, v in acc.items()}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --headline-only --steps 3 --warmup 1 --rounds 1 --prewarm 4` (the transform: `--ntt-only --ntt-variant dft --steps 10`) "
               "(tools/collect_profiles.sh; counter passes serialise the dispatches, so k_acc_tasks does not share the caches with the next step's sort here); "
               "KB per dispatch as reported; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section) -- "
               "calibrated for 16-B-per-lane streams, an upper estimate for k_acc_tasks' 8-B gathers")
def per_launch(kernel, algorithmic):
    f = out["FETCH_SIZE"].get(kernel, {}).get("mean_KB", 0.0) * 1024
    w = out["WRITE_SIZE"].get(kernel, {}).get("mean_KB", 0.0) * 1024
    return {"fetch_reported": f, "fetch_corrected_x2": 2 * f, "write": w, "total_corrected": 2 * f + w, "algorithmic": algorithmic}
# what bench.py's roofline.traffic reads (2^20 G1 pairs: 96 B per pair; 2^22 Fr elements: 32 B read + 32 B written per step)
out["k_acc_tasks_traffic_bytes_per_launch"] = per_launch("k_acc_tasks", 96 << 20)
for k in list(out["FETCH_SIZE"]):                # the steps of a 2^22 transform: 32 B read + 32 B written per element each (+ 36 B of twiddle table in a column step)
    if k.startswith("k_ntt_tile") and k.endswith(f"/grid{1 << 20}"):      # (steps A and B of a three-step plan can share a shape: one mean over both)
        out[f"{k}_traffic_bytes_per_launch"] = per_launch(k, 64 << 22)
# the whole 2^22 transform: every launch of its steps, per transform (the step kernels' launch counts differ when two steps share a shape)
ntt_keys = [k for k in out["FETCH_SIZE"] if k.startswith("k_ntt_tile") and k.endswith(f"/grid{1 << 20}")]
if ntt_keys:
    transforms = min(out["FETCH_SIZE"][k]["dispatches"] for k in ntt_keys)
    f = sum(out["FETCH_SIZE"][k]["mean_KB"] * 1024 * out["FETCH_SIZE"][k]["dispatches"] for k in ntt_keys) / transforms
    w = sum(out["WRITE_SIZE"].get(k, {}).get("mean_KB", 0.0) * 1024 * out["WRITE_SIZE"].get(k, {}).get("dispatches", 0) for k in ntt_keys) / transforms
    out["ntt_traffic_bytes_per_transform"] = {"fetch_reported": f, "fetch_corrected_x2": 2 * f, "write": w, "total_corrected": 2 * f + w, "algorithmic": 64 << 22,
                                              "transforms": transforms}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k in out["FETCH_SIZE"]:
    print(f"{k:32s} fetch {out['FETCH_SIZE'][k]['mean_KB']/1024:9.2f} MB  write {out['WRITE_SIZE'].get(k, {}).get('mean_KB', 0)/1024:9.2f} MB  vgpr {out['FETCH_SIZE'][k]['vgpr']} lds {out['FETCH_SIZE'][k]['lds_bytes']}")
