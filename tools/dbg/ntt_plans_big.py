import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for rep in range(2):
    for tile in ("10", "11", "12"):
        env = dict(os.environ, KG_NTT_STEPS="3", KG_NTT_TILE=tile)
        r = subprocess.run([sys.executable, os.path.join(here, "ntt_sizes.py"), "22", "23", "24", "25"], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("ntt")]
        print(f"steps=3 tile={tile}: " + " | ".join(l.split(":")[1].strip().split("  ")[0] + "@" + l.split(":")[0][4:] for l in lines), flush=True)
