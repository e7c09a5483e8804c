"""Forward NTT time per plan variant (KG_NTT_STEPS / KG_NTT_TILE are read once per process, so every variant is a child):
    python tools/dbg/ntt_plans.py 18 20 22"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
sizes = sys.argv[1:] or ["18", "20", "22"]
for steps in ("0", "3"):
    for tile in ("0", "10", "11", "12"):
        env = dict(os.environ, KG_NTT_STEPS=steps, KG_NTT_TILE=tile)
        r = subprocess.run([sys.executable, os.path.join(here, "ntt_sizes.py")] + sizes, env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("ntt")]
        print(f"steps={steps} tile={tile}: " + " | ".join(l.split(":")[1].strip().split("  ")[0] + "@" + l.split(":")[0][4:] for l in lines) + ("" if r.returncode == 0 else f"  rc={r.returncode} {r.stderr[-200:]}"), flush=True)
