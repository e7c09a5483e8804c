#!/bin/bash
# round 6, first GPU call: the restructured bench line (size, legs), the multi-rank selftest, the short-call baseline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_first; rm -rf "$O"; mkdir -p "$O"
timeout -s KILL 600 python3 bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"
tail -c 300 "$O/bench.err"
timeout -s KILL 200 python3 bench.py --small-only > "$O/small.json" 2> "$O/small.err"; cat "$O/small.json"
timeout -s KILL 900 python3 -m pytest tests/test_gpu_bench_multirank.py -x -q > "$O/multirank.txt" 2>&1; tail -5 "$O/multirank.txt"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_first/bench.json") if l.startswith("{")][-1])
print(json.dumps(d["summary"], indent=0))
PY
