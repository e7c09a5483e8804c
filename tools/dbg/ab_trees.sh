#!/bin/bash
# A/B of two source trees on one box: the round-4 tree (build/r04, git archive of the round's last commit, built in place) against this one --
# the Groth16 legs of each tree's own bench.py, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_trees
for r in 1 2 3; do
  for t in r04 cur; do
    d=$GRAFT_REPO_ROOT; [ $t = r04 ] && d=$GRAFT_REPO_ROOT/build/r04
    ( cd $d; timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-nova --no-skew --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/ab_trees/${t}_$r.json 2> $GRAFT_REPO_ROOT/gpurun_out/ab_trees/${t}_$r.err )
    python - <<PY
import json
l=[json.loads(x) for x in open("$GRAFT_REPO_ROOT/gpurun_out/ab_trees/${t}_$r.json") if x.startswith("{")][-1]
g=l["groth16"]; w=g.get("window_tables",{})
print("$t $r step %.3f blocking %.3f | g16 %.3f / %.3f  tables %.3f / %.3f" % (l["ms_per_step"], l["blocking_ms"], g["ms_per_proof"], g["ms_per_proof_blocking"], w.get("ms_per_proof",0), w.get("ms_per_proof_blocking",0)))
PY
  done
done
