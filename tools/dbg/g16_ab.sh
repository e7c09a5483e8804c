#!/bin/bash
# Groth16 2^18 legs of bench.py, alternating an environment knob off / on:  bash tools/dbg/g16_ab.sh KG_G2_PAIR_ACC=1
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for v in "" "$1"; do
    echo "[$v] $(env $v python3 bench.py --no-cpu-baseline --no-ntt --no-nova --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g = d['groth16']; w = g['window_tables']
print('plain', round(g['ms_per_proof'], 3), round(g['ms_per_proof_blocking'], 3), 'tables', round(w['ms_per_proof'], 3), round(w['ms_per_proof_blocking'], 3), 'match', g['pipelined_matches_blocking'], w['proofs_match'])")"
  done
done
