#!/bin/bash
mkdir -p gpurun_out/r4ab
for r in 1 2 3; do for pl in 0 1; do
KG_QUEUE_PLACEMENT=$pl python bench.py --no-cpu-baseline --no-nova --no-ntt --no-skew --no-groth16 > gpurun_out/r4ab/pl${pl}_$r.json 2>/dev/null
done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4ab/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['blocking_ms'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()})
PY
