#!/bin/bash
# bench.py (MSM + Groth16 legs) under every build/exp/libkg_*.so, alternating rounds on one box:  bash tools/dbg/ab_bench.sh out_dir [rounds]
cd "$(dirname "$0")/../.."
O=${1:-gpurun_out/ab}; mkdir -p $O
for rep in $(seq 1 ${2:-3}); do
  for so in build/exp/libkg_*.so; do
    n=${so##*/libkg_}; n=${n%.so}
    KG_LIB_PATH=$PWD/$so timeout -s KILL 300 python3 bench.py --no-cpu-baseline --no-nova --no-ntt --no-skew > $O/${n}_$rep.json 2> $O/${n}_$rep.err
  done
done
python3 - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); g = d['groth16']; t = g.get('window_tables', {})
        print(f.split('/')[-1], 'msm step', round(d['ms_per_step'], 3), 'blocking', round(d['blocking_ms'], 3), 'g16', round(g['ms_per_proof'], 3), round(g['ms_per_proof_blocking'], 3),
              'tables', round(t.get('ms_per_proof', 0), 3), round(t.get('ms_per_proof_blocking', 0), 3), {k: round(v, 3) for k, v in d['phases_ms_per_step'].items()})
    except Exception as e: print(f, e)
PY
