"""One-line summary of a bench.py JSON line (stdin or a file):  python3 bench.py ... | python3 tools/dbg/bench_line.py"""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
try: d = json.loads(txt)
except ValueError: d = json.loads(txt.strip().splitlines()[-1])
out = ["step", round(d["ms_per_step"], 3), "blocking", round(d.get("blocking_ms", 0), 3)]
g = d.get("groth16")
if g:
    t = g.get("window_tables", {})
    out += ["g16", round(g["ms_per_proof"], 3), round(g["ms_per_proof_blocking"], 3), "tables", round(t.get("ms_per_proof", 0), 3), round(t.get("ms_per_proof_blocking", 0), 3)]
n = d.get("nova_commit")
if n:
    out += ["nova", round(n["g1_fr"]["ms_per_commit"], 2), round(n["grumpkin_fq"]["ms_per_commit"], 2), "unit", round(n["rank_unit"]["blocking_ms_per_commit"], 3)]
if d.get("ntt"): out += ["ntt", round(d["ntt"]["ms"], 4)]
print(*out)
