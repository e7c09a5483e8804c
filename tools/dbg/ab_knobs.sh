#!/bin/bash
# bench.py (MSM + Groth16 legs) under a list of environment settings, alternating rounds on one box:
#   bash tools/dbg/ab_knobs.sh out_dir rounds "NAME=VAL ..." "NAME=VAL ..."      ("-" = no setting)
cd "$(dirname "$0")/../.."
O=$1; R=$2; shift 2; mkdir -p $O
for rep in $(seq 1 $R); do
  i=0
  for kn in "$@"; do
    i=$((i+1))
    if [ "$kn" = "-" ]; then e=""; else e="$kn"; fi
    env $e timeout -s KILL 300 python3 bench.py --no-cpu-baseline --no-nova --no-ntt --no-skew > $O/k${i}_$rep.json 2> $O/k${i}_$rep.err
  done
done
python3 - "$O" "$@" <<'PY'
import json, glob, sys
names = sys.argv[2:]
for f in sorted(glob.glob(sys.argv[1] + '/k*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); g = d['groth16']; t = g.get('window_tables', {})
        k = int(f.split('/')[-1][1:].split('_')[0]) - 1
        print(f"{names[k]:28s}", 'msm step', round(d['ms_per_step'], 3), 'blocking', round(d['blocking_ms'], 3), 'g16', round(g['ms_per_proof'], 3), round(g['ms_per_proof_blocking'], 3),
              'tables', round(t.get('ms_per_proof', 0), 3), round(t.get('ms_per_proof_blocking', 0), 3))
    except Exception as e: print(f, e)
PY
