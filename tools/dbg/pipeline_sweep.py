"""MSM 2^20 step time, phase times and Groth16 ms per proof against the pipeline depth / input semantics:
python tools/dbg/pipeline_sweep.py d2 d3 d4 d3o   (dN: N steps in flight; trailing o: stream-ordered inputs)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for s in sys.argv[1:]:
    extra = []
    if s.endswith("o"):
        s, extra = s[:-1], ["--stream-ordered-inputs"]
    if "d" in s:
        s, dd = s.split("d")
        extra += ["--depth", dd]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(s, "FAILED", r.stderr[-2000:])
        continue
    ph = {k: round(v, 3) for k, v in d["phases_ms_per_step"].items()}
    g = d.get("groth16", {})
    print(f"{' '.join(extra)}: step {d['ms_per_step']:.3f} ms  reg {d['registered_bases']['ms_per_step']:.3f}  ntt {d['ntt']['ms']:.3f}  "
          f"g16 {g.get('ms_per_proof', 0):.3f} / {g.get('ms_per_proof_blocking', 0):.3f} ok={g.get('pipelined_matches_blocking')}  {ph}", flush=True)
